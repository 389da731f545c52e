"""Regenerates tests/golden/oracle_orders.json from the repo's OWN oracle (oracle/nnet_oracle.c).

These vectors do not come from the reference (no JVM in the build image; the reference ships no
fixtures) - they freeze the oracle's current behaviour so that an accidental change of the
restatement, of the synthetic generator or of the build flags shows up as a diff.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nnet_oracle as O  # noqa: E402

out = {"generator": "oracle/nnet_oracle.c via tests/golden/make_golden.py", "cases": []}
for n in (4, 5, 6, 9, 17, 64, 200):
    for seed in (1, 2):
        for dist in ("uniform53", "dec4"):
            D = O.synth(n, seed, dist)
            order, ev, se = O.run(D)
            traj = ev[["m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id"]].tobytes()
            out["cases"].append({
                "n": n, "seed": seed, "dist": dist,
                "matrix_sha256": hashlib.sha256(D.tobytes()).hexdigest(),
                "order": order.tolist() if n <= 64 else None,
                "order_sha256": hashlib.sha256(order.tobytes()).hexdigest(),
                "trajectory_sha256": hashlib.sha256(traj).hexdigest(),
                "best_bits_sha256": hashlib.sha256(ev["best"].tobytes()).hexdigest(),
                "sum_entries": int(se),
            })
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_orders.json"), "w"), indent=1)
print(len(out["cases"]), "cases written")
