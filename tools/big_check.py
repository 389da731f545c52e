"""Screened vs plain-fp64 runs at awkward / large sizes must give the same trajectory."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
a = fa.api()
for n, seed, dist in [(10001, 5, "uniform53"), (12289, 6, "dec4"), (40000, 7, "uniform53")]:
    runs = []
    for disable in (False, True):
        with Handle(a, n, record_events=True, disable_screen=disable) as h:
            h.synth(seed, dist)
            order, st = h.run()
            runs.append((order, h.events(), st))
    (o0, e0, s0), (o1, e1, s1) = runs
    same = (o0 == o1).all() and all((e0[f] == e1[f]).all() for f in ("cx_id", "cy_id", "x_id", "y_id", "kind", "u_id")) \
        and (e0["best"].view(np.int64) == e1["best"].view(np.int64)).all()
    ok = sorted(o0[1:].tolist()) == list(range(1, n + 1))
    print(f"n={n} {dist}: same={same} perm={ok} screened {s0.t_total_s:.2f}s ({s0.n_screen_events} ev, "
          f"{s0.n_rescan_units / max(s0.n_screen_events, 1):.1f} units/ev, rx exact {s0.n_rx_exact}) plain {s1.t_total_s:.2f}s", flush=True)
