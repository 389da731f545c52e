import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
np.set_printoptions(precision=5, linewidth=200)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D = O.synth(n, 1)
st = O.Stepper(D)
h = Handle(fa.api(), n)
h.set_matrix(D); h.begin()
print("init Sx gpu", h.nodes()[2], "oracle", st.nodes()[3])
for k in range(3):
    eo = st.step(); eg = h.step()
    if eo is None: break
    print("event", k, eo.key(), eg.key())
    ids, dist, nbr, sx = st.nodes(); gi, gn, gs = h.nodes()
    print(" oracle ids", ids, "nbr", nbr, "Sx", sx)
    print(" gpu    ids", gi, "nbr", gn, "Sx", gs)
    live = st.matrix()[np.ix_(dist, dist)]
    gl = h.live_matrix()
    print(" live matrix equal:", (live.view(np.int64) == gl.view(np.int64)).all())
    if not (live.view(np.int64) == gl.view(np.int64)).all():
        print(live); print(gl)
