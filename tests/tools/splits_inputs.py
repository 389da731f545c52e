"""Development aid: the split-weight solver on the input classes of tests/inputs.py (tree-like distances have far more
positive splits than random ones).  usage: tests/tools/splits_inputs.py n [class:seed ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from oracle import nnet_oracle as O
from oracle import csw_oracle as W
from common import live_to_fast
import inputs

n = int(sys.argv[1])
cases = [c.split(":") for c in sys.argv[2:]] or [["uniform53", "1"], ["tree", "5"], ["treenoise", "6"]]
for dist, seed in cases:
    D = inputs.make(n, dist, int(seed), O)
    order = fa.canonical_order(D)
    t = time.time(); w, st = fa.split_weights(D, order); dt = time.time() - t
    d = W.setup_d(D, order); x = live_to_fast(n, w)
    g = W.calculate_atx(n, W.calculate_ab(n, x) - d); scale = np.abs(W.calculate_atx(n, d)).max()
    pos = x > 0
    viol = max(float(-x.min()), float(np.abs(g[pos]).max(initial=0.0) / scale), float((-g[~pos]).max(initial=0.0) / scale))
    print(f"n={n} {dist}:{seed}: weights {dt:.2f} s method={st['method']} steps={st['outer_iterations']} rebuilds={st['refactorizations']} "
          f"splits>1e-6={st['nsplits']} positive={int(pos.sum())} kkt={viol:.2e}", flush=True)
