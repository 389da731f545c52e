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
    print(f"relaxed n={n} total={st.t_total_s:.3f}s events={st.n_events} relaxed_events={st.n_relaxed_events} "
          f"plain_launches={st.plain_launches} us_per_relaxed_event={(st.t_agglom_s) / max(st.n_events, 1) * 1e6:.1f}", flush=True)
