import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
a = fa.api()
rng = np.random.default_rng(1)
for m in (16384, 32768):
    v = rng.random(m) + 2.0**-10
    for ept in (32,):
        out = C.c_double(); st = (C.c_int32 * 4)()
        for rep in range(3):
            a.check(a.test_chain_sum(0, v.ctypes.data_as(C.POINTER(C.c_double)), m, 1, ept, C.byref(out), st))
