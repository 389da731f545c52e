"""Development aid: where does a run on a matrix with negative entries spend its time? (per-kernel HIP-event times)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
import inputs
K = ["k_scan", "k_screen", "k_track", "k_decide", "k_update", "k_emit", "k_resolve", "k_finalize"]
a = fa.api()
n = int(sys.argv[1])
for name, dist, kw in [("neg", "neg", {}), ("uniform53, screening off", "uniform53", {"disable_screen": True})]:
    with Handle(a, n, **kw) as h:
        a.set_scan_timing(h._h, 2)
        h.set_matrix(inputs.make(n, dist, 1, O))
        order, st = h.run()
        ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
        a.get_kernel_times(h._h, ms, cnt)
        print(f"{name} n={n}: {st.t_total_s:.3f} s; " + " ".join(f"{K[c]}: {cnt[c]} x {ms[c] * 1e3 / max(cnt[c], 1):.1f} us" for c in range(8) if cnt[c]), flush=True)
