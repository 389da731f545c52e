"""Development aid: checks the lookahead-window invariant on the CPU oracle's trajectory.

For a matrix without negative entries, a cluster pair (P, Q) whose clusters are not involved in
events t .. t+k-1 satisfies   Q_{t+k}(P,Q) >= (c_t - 2 - K) * D(P,Q) - S_P(t) - S_Q(t)   (k <= K):
D(P,Q) is unchanged, the coefficient only shrinks, and the row sums of uninvolved clusters never grow.
This script replays the oracle event by event, asserts that bound for every pair, and simulates
the window policy (threshold Theta = min Q + W at a base scan) to report list sizes and window
lifetimes.  Usage: python tools/lookahead_sim.py n K W [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nnet_oracle  # noqa: E402


def clusters(st):
    """-> list of (key, [distIDs], Sx) for the live clusters; key identifies the cluster."""
    ids, dist, nbr, sx = st.nodes()
    by_id = {int(i): k for k, i in enumerate(ids)}
    out = []
    for k, i in enumerate(ids):
        i = int(i)
        nb = int(nbr[k])
        if nb == 0:
            out.append(((i, 0), [int(dist[k])], float(sx[k])))
        elif nb > i:
            out.append(((i, nb), [int(dist[k]), int(dist[by_id[nb]])], float(sx[k])))
    return out


def qmatrix(st, cl, coef):
    D = st.matrix()
    c = len(cl)
    Dc = np.zeros((c, c))
    S = np.array([x[2] for x in cl])
    # cluster distance = mean of the cross entries
    idx = [x[1] for x in cl]
    first = np.array([i[0] for i in idx])
    second = np.array([i[-1] for i in idx])
    Dc = (D[np.ix_(first, first)] + D[np.ix_(first, second)] + D[np.ix_(second, first)] + D[np.ix_(second, second)]) / 4.0
    Q = coef * Dc - S[:, None] - S[None, :]
    return Q, Dc, S


def main():
    n = int(sys.argv[1]); K = int(sys.argv[2]); W = float(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    D = nnet_oracle.synth(n, seed)
    st = nnet_oracle.Stepper(D)
    base = None      # dict: keys, LB matrix, theta, tracked set, k
    nbase = 0; sizes = []; lifetimes = []; worst = np.inf
    ev_i = 0
    while st.num_active > 4:
        cl = clusters(st)
        c = len(cl)
        keys = [x[0] for x in cl]
        Q, Dc, S = qmatrix(st, cl, c - 2.0)
        iu = np.triu_indices(c, 1)
        qmin = Q[iu].min()
        need = base is None or base["k"] >= K
        if not need:
            # invariant check for pairs of clusters that both exist since the base scan
            pos = {k: i for i, k in enumerate(keys)}
            old = [(pos[k], j) for j, k in enumerate(base["keys"]) if k in pos]
            if len(old) > 1:
                a = np.array([o[0] for o in old]); b = np.array([o[1] for o in old])
                qq = Q[np.ix_(a, a)]; lb = base["LB"][np.ix_(b, b)]
                slack = (qq - lb)[np.triu_indices(len(a), 1)]
                worst = min(worst, slack.min())
                assert slack.min() > -1e-6, ("bound violated", slack.min())
                # tracked minimum: tracked old pairs + everything that involves a fresh cluster
                tr = base["tracked"][np.ix_(b, b)]
                qt = np.where(tr, qq, np.inf)
                np.fill_diagonal(qt, np.inf)
                m_old = qt.min()
            else:
                m_old = np.inf
            fresh = [i for i, k in enumerate(keys) if k not in base["keyset"]]
            m_fresh = np.inf
            for i in fresh:
                row = np.delete(Q[i], i)
                m_fresh = min(m_fresh, row.min())
            M = min(m_old, m_fresh)
            if M <= base["theta"]:
                assert abs(M - qmin) < 1e-9, ("tracked min differs", M, qmin)
            else:
                need = True
        if need:
            if base is not None:
                lifetimes.append(base["k"])
            LB = (c - 2.0 - K) * Dc - S[:, None] - S[None, :]
            theta = qmin + W
            tracked = LB <= theta
            np.fill_diagonal(tracked, False)
            base = {"keys": keys, "keyset": set(keys), "LB": LB, "theta": theta, "tracked": tracked, "k": 0}
            sizes.append(int(tracked.sum() // 2))
            nbase += 1
        base["k"] += 1
        if st.step() is None:
            break
        ev_i += 1
    print(f"n={n} K={K} W={W}: events={ev_i} base scans={nbase} mean list={np.mean(sizes):.1f} max list={max(sizes)} "
          f"mean lifetime={np.mean(lifetimes) if lifetimes else 0:.1f} smallest slack of the bound={worst:.3g}")


if __name__ == "__main__":
    main()
