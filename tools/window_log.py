"""Development aid: per-base-scan log of the lookahead windows (fnn_debug_window_log)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
a = fa.api()
a._fn("debug_window_log", C.c_int64, [C.c_void_p, C.POINTER(C.c_double), C.c_int64])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
with Handle(a, n) as h:
    h.synth(1, "uniform53")
    order, st = h.run()
    buf = np.zeros((8192, 5))
    k = a.debug_window_log(h._h, buf.ctypes.data_as(C.POINTER(C.c_double)), 8192)
print(f"n={n} total={st.t_total_s:.3f}s base_scans={st.n_base_scans} hits={st.n_window_hits} fails={st.n_window_fails} records={k}")
step = max(1, k // 120)
print("event m W pairs served_by_previous_window")
for r in buf[:k:step]:
    print(int(r[0]), int(r[1]), f"{r[2]:.2f}", int(r[3]), int(r[4]))
life = buf[:k, 4]
print("served events per window: mean", life[life >= 0].mean(), "median", np.median(life[life >= 0]))
