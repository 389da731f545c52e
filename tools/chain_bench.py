"""Development aid: duration of the exact chain-sum kernel (FNN_CHAIN_TIME=1; FNN_CHAIN_STOP=1..3 stops
after the loads + prefix / the automata / the segmented scan to show where the time goes)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
a = fa.api()
os.environ["FNN_CHAIN_TIME"] = "1"
rng = np.random.default_rng(1)
for m in (4096, 16384, 32768):
    v = rng.random(m) + 2.0**-10
    for stop in (1, 2, 3, 4, 5, 0):
        os.environ["FNN_CHAIN_STOP"] = str(stop)
        out = C.c_double(); st = (C.c_int32 * 4)()
        print(f"m={m} stop_after={stop}:", flush=True)
        a.check(a.test_chain_sum(0, v.ctypes.data_as(C.POINTER(C.c_double)), m, 1, 32, C.byref(out), st))
        if stop == 0:
            print(f"[fnn] stats: runs={st[0]} mixed={st[1]} run_fail={st[2]} thread_fail={st[3]}", flush=True)
