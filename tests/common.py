"""Shared helpers: event-by-event comparison of an engine (HIP or emulation) with the oracle."""
import numpy as np

from fastneighbornet_amd._capi import Handle


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def compare_trajectory(api, oracle, D, deep=True, deep_every=1, **hkw):
    """Step engine and oracle together; every event record must agree exactly; with
    `deep`, node ids / partners / Sx bits / the whole live matrix are compared too.
    (relaxed_seed / relaxed_min_active in hkw put both sides into the Relaxed mode.)"""
    n = D.shape[0]
    st = oracle.Stepper(D, relaxed_seed=hkw.get("relaxed_seed"), relaxed_min_active=hkw.get("relaxed_min_active", 0))
    h = Handle(api, n, **hkw)
    try:
        h.set_matrix(D)
        h.begin()
        k = 0
        while True:
            eo = st.step()
            eg = h.step()
            if eo is None or eg is None:
                assert eo is None and eg is None, (k, eo, eg)
                break
            assert eo.key() == eg.key(), (k, eo.key(), eg.key())
            assert bits([eo.best])[0] == bits([eg.best])[0], (k, eo.best, eg.best)
            assert eo.entries == eg.entries
            if deep and k % deep_every == 0:
                ids, dist, nbr, sx = st.nodes()
                gi, gn, gs = h.nodes()
                assert (ids == gi).all() and (nbr == gn).all(), (k, ids, gi, nbr, gn)
                assert (bits(sx) == bits(gs)).all(), (k, sx, gs)
                if (dist >= 0).all():
                    live = st.matrix()[np.ix_(dist, dist)]
                    assert (bits(live) == bits(h.live_matrix())).all(), k
            k += 1
        o1 = st.expand()
        o2 = h.finish()
        assert (o1 == o2).all(), (o1, o2)
        return k, o2
    finally:
        h.close()
        st.close()


def check_order(order, n):
    """Structural invariants of SURVEY.md section 4."""
    assert len(order) == n + 1
    assert order[0] == 0
    if n >= 1:
        assert order[1] == 1
    assert sorted(order[1:].tolist()) == list(range(1, n + 1))


def tree_metric(n, seed):
    """Additive tree metric with dyadic branch lengths (path sums exact -> exact ties of the Q criterion)."""
    r = np.random.default_rng(seed)
    parent = [-1]; blen = [0.0]
    leaves = [0]
    while len(leaves) < n:                      # split a random leaf into two
        v = leaves.pop(int(r.integers(len(leaves))))
        for _ in range(2):
            parent.append(v); blen.append(float(r.integers(1, 64)) / 64.0); leaves.append(len(parent) - 1)
    def path(v):
        out = []
        while v >= 0:
            out.append(v); v = parent[v]
        return out
    P = [path(v) for v in leaves]
    D = np.zeros((n, n))
    for i in range(n):
        si = set(P[i]); di = {v: sum(blen[u] for u in P[i][:k]) for k, v in enumerate(P[i])}
        for j in range(i + 1, n):
            dj = 0.0
            for v in P[j]:
                if v in si:
                    D[i, j] = D[j, i] = di[v] + dj
                    break
                dj += blen[v]
    return D


def live_to_fast(n, live):
    """Split weights in the live path's index order (FastNN.java:405-419) -> the fast algorithm's packed upper triangle
    (SURVEY.md App. D), one vector operation per row (the element-wise form is tests/test_split_weights.py: fast_x)."""
    x = np.zeros(n * (n - 1) // 2)
    k = 0
    for i in range(n - 1):
        j = np.arange(i + 1, n, dtype=np.int64)
        if i >= 1:
            fi, fj = i - 1, j - 1
        else:
            fi, fj = j - 1, n - 1
        x[(2 * n - fi - 3) * fi // 2 + fj - 1] = live[k:k + len(j)]
        k += len(j)
    return x
