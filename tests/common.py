"""Shared helpers: event-by-event comparison of an engine (HIP or emulation) with the oracle."""
import numpy as np

from fastneighbornet_amd._capi import Handle


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def compare_trajectory(api, oracle, D, deep=True, deep_every=1, **hkw):
    """Step engine and oracle together; every event record must agree exactly; with
    `deep`, node ids / partners / Sx bits / the whole live matrix are compared too."""
    n = D.shape[0]
    st = oracle.Stepper(D)
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
