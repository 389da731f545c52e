"""Development aid: run time and window statistics on inputs other than the bench's uniform matrices:
exact additive tree metrics (dyadic branch lengths: exact ties of the Q criterion everywhere), the same
with noise, and the 4-decimal generator.  usage: tools/other_inputs.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle


def tree_metric(n, seed, dyadic=True):
    r = np.random.default_rng(seed)
    D = np.zeros((n, n))
    stack = [(0, n, 0.0)]
    while stack:
        a, b, h = stack.pop()           # h = length of the path from the root to this node
        if b - a == 1:
            continue
        c = int(r.integers(a + 1, b))
        la = float(r.integers(1, 64)) / 64.0 if dyadic else float(r.random()) + 0.01
        lb = float(r.integers(1, 64)) / 64.0 if dyadic else float(r.random()) + 0.01
        stack.append((a, c, h + la))
        stack.append((c, b, h + lb))
        D[a:c, c:b] -= 2.0 * h          # -2 h(lca); the leaf depths are added below
    depth = np.zeros(n)
    stack = [(0, n, 0.0)]
    r = np.random.default_rng(seed)
    while stack:
        a, b, h = stack.pop()
        if b - a == 1:
            depth[a] = h
            continue
        c = int(r.integers(a + 1, b))
        la = float(r.integers(1, 64)) / 64.0 if dyadic else float(r.random()) + 0.01
        lb = float(r.integers(1, 64)) / 64.0 if dyadic else float(r.random()) + 0.01
        stack.append((a, c, h + la))
        stack.append((c, b, h + lb))
    iu = np.triu_indices(n, 1)
    D[iu] += depth[iu[0]] + depth[iu[1]]
    D = np.triu(D, 1)
    D = D + D.T
    p = r.permutation(n)
    return np.ascontiguousarray(D[np.ix_(p, p)])


def run(name, n, D=None, synth=None):
    a = fa.api()
    with Handle(a, n) as h:
        if D is not None:
            h.set_matrix(D)
        else:
            h.synth(*synth)
        t = time.time()
        order, st = h.run()
        dt = time.time() - t
    print(f"{name} n={n}: {st.t_total_s:.3f} s (wall {dt:.3f}) base_scans={st.n_base_scans} window_hits={st.n_window_hits} "
          f"window_fails={st.n_window_fails} stalled={st.n_stalled_events} rx_exact={st.n_rx_exact} rx_certified={st.n_rx_certified} "
          f"exact_sweeps={st.n_sweeps_exact}", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    run("uniform53", n, synth=(1, "uniform53"))
    run("dec4", n, synth=(1, "dec4"))
    T = tree_metric(n, 5)
    assert (T >= 0).all() and (T == T.T).all()
    run("tree (dyadic, exact ties)", n, D=T)
    rng = np.random.default_rng(3)
    N = np.triu(rng.random((n, n)) * 0.05, 1)
    run("tree + 5 % noise", n, D=tree_metric(n, 6, dyadic=False) + N + N.T)
