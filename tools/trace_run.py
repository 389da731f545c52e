"""One whole run with the event log on; saves the per-event records (kind, m, c) next to a rocprofv3
kernel trace of the same process so that tools/chain_table.py can join kernels to events.
usage: trace_run.py <n> <out.npz>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle

n = int(sys.argv[1])
a = fa.api()
with Handle(a, n, record_events=True) as h:
    h.synth(1, "uniform53")
    order, st = h.run()
    ev = h.events()
np.savez_compressed(sys.argv[2], kind=ev["kind"], m=ev["m_before"], c=ev["c_before"],
                    total_s=st.t_total_s, agglom_s=st.t_agglom_s, base_scans=st.n_base_scans,
                    stalled=st.n_stalled_events)
print(f"n={n} total={st.t_total_s:.3f}s events={st.n_events} base_scans={st.n_base_scans} stalled={st.n_stalled_events}")
