"""Oracle goldens at BASELINE.json's sizes (n = 4096 / 16384 / 32768), generated in the BUILD container.

Runs the repo's own CPU oracle (oracle/nnet_oracle.c, OpenMP scan reduced on the total order
(Q, i, j), which returns the serial scan's pair by construction and by tests/test_oracle.py) on the
synthetic matrices of SURVEY.md 8(d) and freezes, per case,
  * tests/golden/oracle_big.json: sha256 of the order, of the trajectory, of the scan minima's bit
    patterns, sum of E_t, wall-clock seconds and thread count of the generating run;
  * tests/golden/big_<n>_<dist>_s<seed>.npz: the order and the whole per-event trajectory
    (compressed, < 1 MB), so that a GPU mismatch can be located at its first diverging event.
These vectors do NOT come from the reference (no JVM in the image, the reference ships no
fixtures): PARITY UNPINNED still holds; what they pin is the GPU engine at the headline sizes
against the restatement that was read against the Java line by line.

n = 32768 reads 1.13e13 matrix entries: about an hour on 8 cores, 16 GiB of memory.
Usage (repo root):  python tests/golden/make_golden_big.py [n:dist:seed ...]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nnet_oracle as O  # noqa: E402
import inputs  # noqa: E402  (tests/inputs.py: the input classes)

GOLD = os.path.join(ROOT, "tests", "golden")
JSON = os.path.join(GOLD, "oracle_big.json")
TRAJ_FIELDS = ["m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id"]
DEFAULT = ["4096:uniform53:1", "4096:dec4:1", "16384:uniform53:1", "32768:uniform53:1"]
# round 3: the other primary seeds of BASELINE.md section 3 and the input classes of tests/inputs.py
ROUND3 = ["4096:uniform53:2", "4096:uniform53:3", "4096:tree:5", "4096:treenoise:6", "4096:neg:1",
          "8192:tree:7", "8192:treenoise:8", "8192:neg:2", "16384:uniform53:2", "16384:uniform53:3", "16384:tree:9", "16384:treenoise:10", "16384:neg:3"]
sha_big = inputs.sha_big


def main(argv):
    threads = int(os.environ.get("GOLDEN_THREADS", os.cpu_count() or 1))
    doc = json.load(open(JSON)) if os.path.exists(JSON) else {
        "generator": "oracle/nnet_oracle.c (OpenMP scan) via tests/golden/make_golden_big.py", "cases": []}
    if argv == ["round3"]:
        argv = ROUND3
    for spec in (argv or DEFAULT):
        n, dist, seed = spec.split(":")
        n, seed = int(n), int(seed)
        D = inputs.make(n, dist, seed, O)
        t0 = time.time()
        order, ev, se = O.run(D, threads=threads)
        dt = time.time() - t0
        traj = np.ascontiguousarray(np.stack([ev[f] for f in TRAJ_FIELDS], axis=1))
        case = {
            "n": n, "seed": seed, "dist": dist,
            "matrix_sha256": sha_big(D),
            "order_sha256": hashlib.sha256(order.tobytes()).hexdigest(),
            # same byte layout as make_golden.py: the 8 int32 fields of every event, event-major
            "trajectory_sha256": hashlib.sha256(traj.tobytes()).hexdigest(),
            "best_bits_sha256": hashlib.sha256(np.ascontiguousarray(ev["best"]).tobytes()).hexdigest(),
            "sum_entries": int(se), "n_events": int(len(ev)),
            "oracle_seconds": round(dt, 2), "oracle_threads": threads,
            "npz": f"big_{n}_{dist}_s{seed}.npz",
        }
        np.savez_compressed(os.path.join(GOLD, case["npz"]), order=order, traj=traj,
                            best_bits=np.ascontiguousarray(ev["best"]).view(np.int64))
        doc["cases"] = [c for c in doc["cases"] if (c["n"], c["seed"], c["dist"]) != (n, seed, dist)] + [case]
        json.dump(doc, open(JSON, "w"), indent=1)
        print(f"n={n} {dist} seed={seed}: {dt:.1f} s on {threads} threads, sum E_t = {se}", flush=True)
        del D


if __name__ == "__main__":
    main(sys.argv[1:])
