"""Development aid (uses the oracle: test infrastructure): the split-weight solver on NEARLY CIRCULAR distances - a circular
metric with 1 % multiplicative noise (tests/inputs.py: circ_noise) - whose optimum has O(n^2) positive splits: more than
the dense factor of the block active-set method holds.  Prints splits against capacity, the route taken and the seconds.
usage: tests/tools/near_circular.py n [n ...] [--noise 0.01] [--density 1.0]     (FNN_SW_LOG=1 for the solver's trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import FnnError
import inputs

args = sys.argv[1:]
noise = float(args[args.index("--noise") + 1]) if "--noise" in args else 0.01
density = float(args[args.index("--density") + 1]) if "--density" in args else 1.0
sizes = [int(a) for a in args if a.isdigit()]
for n in sizes:
    D = inputs.circ_noise(n, 11, noise, density)
    t = time.time(); order = fa.canonical_order(D); t_o = time.time() - t
    t = time.time()
    try:
        w, st = fa.split_weights(D, order, allow_inexact=True)
        dt = time.time() - t
        print(f"n={n} noise={noise} density={density}: order {t_o:.2f} s; weights {dt:.2f} s route={st['method']} certified={st['certified']} "
              f"kkt={st['kkt_violation']:.2e} splits>1e-6={st['nsplits']} ({st['nsplits'] / n:.1f} n, {st['nsplits'] / (n * (n - 1) / 2):.3f} of all) "
              f"capacity={st['capacity']} free_peak={st['free_set_peak']} giveup={st['giveup_reason']} cg_calls={st['cg_calls']} cg_its={st['cg_iterations']}", flush=True)
    except FnnError as e:
        dt = time.time() - t
        print(f"n={n} noise={noise} density={density}: order {t_o:.2f} s; weights FAILED after {dt:.2f} s: {e} | stats {getattr(e, 'stats', None)}", flush=True)
