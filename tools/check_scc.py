"""Flags s_cselect / s_cbranch_scc whose most recent SCC writer is not a compare-like
instruction (workaround detector for the hipcc 7.2 v_cmp -> s_cselect miscompile)."""
import re, sys
scc_writers = re.compile(r"^\s*(s_cmp|s_bitcmp|s_add|s_sub|s_addc|s_subb|s_and|s_or|s_xor|s_not|s_lshl|s_lshr|s_ashr|s_min|s_max|s_abs|s_andn2|s_orn2|s_nand|s_nor|s_xnor|s_bfe|s_mul_i32_SKIP|s_absdiff|s_wqm|s_quadmask|s_bcnt|s_ff|s_flbit|s_cmpk|s_addk|s_mulk_SKIP)")
cmp_like = re.compile(r"^\s*(s_cmp|s_bitcmp|s_cmpk|s_and_b64|s_or_b64|s_andn2_b64|s_and_b32|s_or_b32|s_xor_b64|s_orn2_b64)")
kern = None; last = None; lastline = 0; bad = 0
for i, line in enumerate(open(sys.argv[1]), 1):
    m = re.match(r"^(_Z\w+):", line)
    if m: kern = m.group(1); last = None
    if re.match(r"^\.LBB", line): pass
    s = line.strip()
    if s.startswith("s_cselect") or s.startswith("s_cbranch_scc"):
        if last is None or not cmp_like.match(last):
            print(f"{kern} line {i}: {s}   <- last SCC writer (line {lastline}): {last.strip() if last else None}")
            bad += 1
    if scc_writers.match(line):
        last = line; lastline = i
print("flagged:", bad)
