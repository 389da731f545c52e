"""Development aid: split-weight solver timings on random distances with the engine's own order."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fastneighbornet_amd as fa
from oracle import nnet_oracle as O
for n in [int(a) for a in sys.argv[1:]] or [128, 257, 400]:
    D = O.synth(n, 1)
    t = time.time(); order = fa.canonical_order(D); t_order = time.time() - t
    t = time.time(); w, st = fa.split_weights(D, order, allow_inexact=True); dt = time.time() - t
    print(f"n={n} order {t_order:.2f}s weights {dt:.2f}s (device {st['t_solve_s']:.2f}s) method={st['method']} steps={st['outer_iterations']} "
          f"refactorizations={st['refactorizations']} solves={st['solves']} entered={st['entered']} screened_out={st['screened_out']} departed={st['departed']} "
          f"splits={st['nsplits']} certified={st['certified']} kkt={st['kkt_violation']:.2e} threshold={st['final_threshold_rel']:.1e} set_aside={st['n_set_aside']}", flush=True)
