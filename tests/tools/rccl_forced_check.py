"""One-GPU rehearsal of the RCCL exchange path (FNN_COMM_FORCE=1: 1-rank communicator, the
all-gather still runs every event), with screening forced on so that k_resolve's records are
gathered as they are."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (bench.py order: torch's HIP runtime serves both)
import fastneighbornet_amd as fa
from fastneighbornet_amd import distributed as fd
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
a = fa.api(); n = 3000
D = O.synth(n, 3); o_ref, _, _ = O.run(D, threads=8)
buf = (C.c_uint8 * 128)(); path = fd.rccl_path()
with Handle(a, n) as h:
    a.check(a.comm_unique_id(buf, path.encode() if path else None))
    h.comm_init_rccl(1, 0, bytes(buf), path)
    h.set_matrix(D); order, st = h.run()
print("RCCL_FORCED_OK" if (order == o_ref).all() else "RCCL_FORCED_MISMATCH", st.n_screen_events, st.n_rescan_units)
