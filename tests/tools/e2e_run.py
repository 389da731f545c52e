"""BASELINE config 5 end to end: circular order + circular split weights + Nexus document for n synthetic taxa on one GPU,
with the solver-independent Kuhn-Tucker certificate of the weights computed on the host (the split-weight oracle's
operators: test infrastructure, which is why this script lives under tests/).

usage: tests/tools/e2e_run.py n [seed] [--json out.json] [--no-kkt] [--no-nexus]
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import fastneighbornet_amd as fa
from oracle import csw_oracle as W
from oracle import nnet_oracle as O
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from common import live_to_fast  # noqa: E402


def kkt_violation(D, order, live):
    """largest violation of x >= 0, of g >= 0 on the zero weights and of g = 0 on the positive ones (g = A^T (A x - d)),
    relative to max |A^T d| (tests/test_split_weights.py: kkt_violation)"""
    n = D.shape[0]
    d = W.setup_d(D, order)
    x = live_to_fast(n, live)
    g = W.calculate_atx(n, W.calculate_ab(n, x) - d)
    scale = np.abs(W.calculate_atx(n, d)).max()
    pos = x > 0
    return dict(min_x=float(x.min()), grad_on_positive=float(np.abs(g[pos]).max(initial=0.0) / scale),
                grad_on_zero=float((-g[~pos]).max(initial=0.0) / scale), positive=int(pos.sum()))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]); seed = int(args[1]) if len(args) > 1 else 1
    out = {"n": n, "seed": seed, "dist": "uniform53"}
    t = time.time(); D = O.synth(n, seed); out["t_synth_host_s"] = round(time.time() - t, 3)
    t = time.time(); order = fa.canonical_order(D); out["t_order_s_incl_upload"] = round(time.time() - t, 3)
    print(f"n={n}: order in {out['t_order_s_incl_upload']} s (with the upload of the host matrix)", flush=True)
    t = time.time(); w, st = fa.split_weights(D, order); out["t_weights_wall_s"] = round(time.time() - t, 3)
    out["weights"] = {k: (float(v) if isinstance(v, float) else v) for k, v in st.items()}
    print(f"weights in {out['t_weights_wall_s']} s (device {st['t_solve_s']:.2f} s): {st}", flush=True)
    if "--no-nexus" not in sys.argv:
        host = C.CDLL(os.path.join(os.path.dirname(fa.__file__), "libfastnn_host.so"))
        host.fnnh_write_nexus.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        host.fnnh_write_nexus.restype = C.c_int32
        names = b"".join((f"t{i + 1}".encode()).ljust(256, b"\0") for i in range(n))
        path = os.environ.get("E2E_NEXUS", "/dev/null")
        t = time.time()
        ns = host.fnnh_write_nexus(path.encode(), n, D.ctypes.data_as(C.POINTER(C.c_double)), names,
                                   order.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(C.POINTER(C.c_double)))
        out["t_nexus_s"] = round(time.time() - t, 3); out["nexus_splits"] = int(ns); out["nexus_path"] = path
        print(f"Nexus document ({ns} splits) to {path} in {out['t_nexus_s']} s", flush=True)
    if "--no-kkt" not in sys.argv:
        t = time.time(); out["kkt"] = kkt_violation(D, order, w); out["t_kkt_host_s"] = round(time.time() - t, 3)
        out["kkt"]["violation"] = max(-out["kkt"]["min_x"], out["kkt"]["grad_on_positive"], out["kkt"]["grad_on_zero"])
        print(f"Kuhn-Tucker certificate: {out['kkt']} ({out['t_kkt_host_s']} s on the host)", flush=True)
    out["t_end_to_end_s"] = round(out["t_order_s_incl_upload"] + out["t_weights_wall_s"] + out.get("t_nexus_s", 0.0), 3)
    print(json.dumps(out), flush=True)
    for i, a in enumerate(sys.argv):
        if a == "--json":
            json.dump(out, open(sys.argv[i + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
