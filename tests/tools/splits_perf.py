"""Development aid: timings of the circular split-weight solve (GPU vs the CPU oracle)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fastneighbornet_amd as fa
from oracle import nnet_oracle as O, csw_oracle as W
for n in [int(x) for x in sys.argv[1:]] or [64, 128, 257]:
    D = O.synth(n, 1)
    order = fa.canonical_order(D)
    t = time.time(); got, st = fa.split_weights(D, order); tg = time.time() - t
    t = time.time(); ref, sr = W.split_weights(D, order); tc = time.time() - t
    print(f"n={n} gpu={tg:.2f}s (solve {st['t_solve_s']:.2f}s, outer={st['outer_iterations']} cg_calls={st['cg_calls']} "
          f"cg_iterations={st['cg_iterations']}, {st['t_solve_s'] / max(st['cg_iterations'], 1) * 1e6:.0f} us per CG iteration) "
          f"cpu_oracle={tc:.2f}s (cg_iterations={sr[2]}) max|diff|={np.abs(got - ref).max():.2e} splits={st['nsplits']}", flush=True)
