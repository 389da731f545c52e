"""Development aid: sweep the lookahead-window parameters (environment switches) at one size."""
import os, subprocess, sys, itertools
n = sys.argv[1] if len(sys.argv) > 1 else "32768"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(env):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "quick_perf.py"), n], env=e, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if l.startswith("n="):
            t = l.split("total=")[1].split("s")[0]
            bs = l.split("base_scans=")[1].split()[0]; wf = l.split("window_fails=")[1].split()[0]
            return f"{t} s  base_scans={bs} fails={wf}"
    return out[-200:]
print("default", run({}), flush=True)
for target, pcap in [(16384, 65536), (24576, 65536), (49152, 65536)]:
    print(f"TARGET={target}", run({"FNN_LA_TARGET": str(target)}), flush=True)
for k, kb in [(32, 16), (64, 16), (64, 32), (48, 8), (96, 32)]:
    print(f"K={k} KBASE={kb}", run({"FNN_LA_K": str(k), "FNN_LA_KBASE": str(kb)}), flush=True)
for kd in (512, 2048):
    print(f"KDIV={kd}", run({"FNN_LA_KDIV": str(kd)}), flush=True)
for b in (32, 128):
    print(f"BATCH={b}", run({"FNN_BATCH": str(b)}), flush=True)
for g in (32, 128):
    print(f"TRACK_GRID={g}", run({"FNN_TRACK_GRID": str(g)}), flush=True)
