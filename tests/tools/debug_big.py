"""Development aid: where does a configuration of the engine leave the oracle's golden trajectory?
usage: tests/tools/debug_big.py n dist seed   (runs the default mode, windows off, screening off)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
import inputs

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
F = ["m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id"]


def run(n, dist, seed, env, **hkw):
    for k, v in env.items():
        os.environ[k] = v
    try:
        a = fa.api()
        with Handle(a, n, record_events=True, **hkw) as h:
            if dist in inputs.DEVICE_DISTS:
                h.synth(seed, dist)
            else:
                h.set_matrix(inputs.make(n, dist, seed, O))
            order, st = h.run()
            ev = h.events()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return order, st, ev


def main():
    n, dist, seed = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    c = {(c["n"], c["dist"], c["seed"]): c for c in json.load(open(os.path.join(GOLD, "oracle_big.json")))["cases"]}[(n, dist, seed)]
    z = np.load(os.path.join(GOLD, c["npz"]))
    for name, env, hkw in [("default", {}, {}), ("windows off", {"FNN_LA_K": "-1"}, {}), ("screening off", {}, {"disable_screen": True}),
                           ("exact rx forced", {}, {"force_exact_rx": True}), ("no deferred chain", {"FNN_NO_DEFER": "1"}, {}),
                           ("default again", {}, {}), ("default, third time", {}, {}),
                           ("window events as two launches (FNN_FUSE=0)", {"FNN_FUSE": "0"}, {}),
                           ("fused, ComputeRx helpers off", {"FNN_RX_HELPERS": "0"}, {}),
                           ("two launches, ComputeRx helpers off", {"FNN_FUSE": "0", "FNN_RX_HELPERS": "0"}, {})]:
        try:
            order, st, ev = run(n, dist, seed, env, **hkw)
        except Exception as e:  # noqa: BLE001
            print(f"{name}: {type(e).__name__}: {e}")
            continue
        traj = np.stack([ev[f] for f in F], axis=1)
        best = np.ascontiguousarray(ev["best"]).view(np.int64)
        k = min(len(traj), len(z["traj"]))
        bad = np.nonzero((traj[:k] != z["traj"][:k]).any(axis=1) | (best[:k] != z["best_bits"][:k]))[0]
        print(f"{name}: {'identical' if bad.size == 0 else 'first diverging event %d: gpu %s best %r oracle %s best %r' % (bad[0], traj[bad[0]].tolist(), float(ev['best'][bad[0]]), z['traj'][bad[0]].tolist(), float(z['best_bits'][bad[0]:bad[0]+1].view(np.float64)[0]))}"
              f" | {st.t_total_s:.3f} s base_scans={st.n_base_scans} hits={st.n_window_hits} fails={st.n_window_fails} rx_exact={st.n_rx_exact} sweeps_exact={st.n_sweeps_exact} "
              f"retries={st.n_handover_retries}", flush=True)


if __name__ == "__main__":
    main()
