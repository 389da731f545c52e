"""Quick on-GPU timing of the Relaxed mode (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle

a = fa.api()
for n in [int(x) for x in sys.argv[1:]] or [4096]:
    with Handle(a, n, relaxed_seed=1) as h:
        h.synth(1, "uniform53")
        order, st = h.run()
        import ctypes as C
        tk = (C.c_int64 * 4)()
        a._fn("debug_relaxed_ticks", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)])
        a.debug_relaxed_ticks(h._h, tk)
        if tk[3]:
            print(f"  row minima per relaxed event {tk[3] / max(st.n_relaxed_events, 1):.1f}; us per row minimum: pass {tk[0] / 100 / tk[3]:.2f} "
                  f"finish {tk[1] / 100 / tk[3]:.2f}; search kernel us per event {tk[2] / 100 / max(st.n_relaxed_events, 1):.1f}")
    print(f"relaxed n={n} total={st.t_total_s:.3f}s events={st.n_events} relaxed_events={st.n_relaxed_events} "
          f"plain_launches={st.plain_launches} us_per_relaxed_event={(st.t_agglom_s) / max(st.n_events, 1) * 1e6:.1f}", flush=True)
