"""Per-kernel ISA resources of the two HIP sources (VGPR / SGPR / spills / scratch / LDS / waves per SIMD) from
hipcc -Rpass-analysis=kernel-resource-usage, as a markdown table.   usage: tools/isa_resources.py [out.md]"""
import hashlib, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "fastneighbornet_amd", "csrc", f) for f in ("fnn_hip.hip", "fnn_splits.hip")]


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o) for o in out[: len(names)]]


def main():
    rows = []
    for src in SRC:
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only",
                                "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "o.o"), src], capture_output=True, text=True)
        cur = None
        for line in r.stderr.split("\n"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"src": os.path.basename(src), "name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    keep = [i for i, nm in enumerate(names) if "rocprim" not in nm and "hipcub" not in nm]
    rows, names = [rows[i] for i in keep], [names[i] for i in keep]
    out = ["| kernel | source | VGPR | AGPR | SGPR | VGPR spills | SGPR spills | scratch B/lane | LDS B/block | waves/SIMD |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r, nm in zip(rows, names):
        out.append(f"| `{nm}` | {r['src']} | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('VGPRs Spill')} | {r.get('SGPRs Spill')} | "
                   f"{r.get('ScratchSize [bytes/lane]')} | {r.get('LDS Size [bytes/block]')} | {r.get('Occupancy [waves/SIMD]')} |")
    txt = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)
        doc = {"fnn_hip_sha256": hashlib.sha256(open(SRC[0], "rb").read()).hexdigest(),
               "compiler": subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout.split("\n")[0],
               "kernels": {nm: {"vgpr": int(r.get("VGPRs", 0)), "sgpr": int(r.get("TotalSGPRs", 0)), "vgpr_spills": int(r.get("VGPRs Spill", 0)),
                                "sgpr_spills": int(r.get("SGPRs Spill", 0)), "scratch_bytes_per_lane": int(r.get("ScratchSize [bytes/lane]", 0)),
                                "lds_bytes_per_block": int(r.get("LDS Size [bytes/block]", 0)), "waves_per_simd": int(r.get("Occupancy [waves/SIMD]", 0))}
                           for r, nm in zip(rows, names)}}
        json.dump(doc, open(os.path.splitext(sys.argv[1])[0] + ".json", "w"), indent=1)
    print(txt)


if __name__ == "__main__":
    main()
