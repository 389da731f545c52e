import os, subprocess, sys
root = sys.argv[1]
def run(env, n="32768"):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "quick_perf.py"), n], env=e, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if l.startswith("n=") and "total=" in l:
            t = l.split("total=")[1].split("s")[0]
            bs = l.split("base_scans=")[1].split()[0]; wf = l.split("window_fails=")[1].split()[0]; stl = l.split("stalled=")[1].split()[0]
            return f"{t} s  base_scans={bs} fails={wf} stalled={stl}"
    return out[-200:]
for env in [{}, {"FNN_BATCH": "128"}, {"FNN_BATCH": "256"}, {"FNN_BATCH": "128", "FNN_LA_KDIV": "512"}, {"FNN_BATCH": "128", "FNN_LA_KDIV": "512", "FNN_LA_TARGET": "49152"},
            {"FNN_BATCH": "128", "FNN_LA_KDIV": "768", "FNN_LA_TARGET": "49152"}, {"FNN_BATCH": "192", "FNN_LA_KDIV": "512", "FNN_LA_TARGET": "49152"},
            {"FNN_BATCH": "128", "FNN_LA_KDIV": "384", "FNN_LA_TARGET": "49152"}]:
    print(env, run(env), flush=True)
for env in [{}, {"FNN_BATCH": "128"}, {"FNN_BATCH": "128", "FNN_LA_KDIV": "512", "FNN_LA_TARGET": "49152"}]:
    print("n=16384", env, run(env, "16384"), flush=True)
    print("n=4096", env, run(env, "4096"), flush=True)
