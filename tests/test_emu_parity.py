"""CPU tests of the engine's host logic and per-thread bodies through the emulation
driver (tests/emu): slot layout bookkeeping, micro-op plans, expandNodes.  The
thread bodies are the ones the HIP kernels wrap; running them in reverse and
shuffled order exposes order dependence (= races on the GPU)."""
import numpy as np
import pytest

from common import check_order, compare_trajectory, tree_metric
from fastneighbornet_amd._capi import Handle


# mode = thread order (0 forward, 1 reverse, 2 shuffled) + 3 * interleaving of k_update's bulk
# threads with its special phases (0 bulk first, 1 special first, 2 mixed)
@pytest.mark.parametrize("mode", list(range(9)))
@pytest.mark.parametrize("dist", ["uniform53", "dec4"])
def test_small_sizes_deep(emu_api, oracle, dist, mode):
    emu_api.set_order_mode(mode)
    for n in [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 31, 32, 33, 64, 65]:
        for seed in (1, 2, 3):
            compare_trajectory(emu_api, oracle, oracle.synth(n, seed, dist), deep=True)


@pytest.mark.parametrize("n,seed,dist,mode", [(257, 1, "uniform53", 8), (300, 2, "dec4", 4),
                                              (400, 3, "uniform53", 0)])
def test_medium_sizes(emu_api, oracle, n, seed, dist, mode):
    emu_api.set_order_mode(mode)
    k, order = compare_trajectory(emu_api, oracle, oracle.synth(n, seed, dist), deep=True, deep_every=37)
    check_order(order, n)


def test_tiny_identity(emu_api, oracle):
    # ntax <= 3: identity ordering (NetMakerOriginal.java:133-140)
    emu_api.set_order_mode(0)
    for n in (1, 2, 3):
        with Handle(emu_api, n) as h:
            h.set_matrix(oracle.synth(n, 1))
            order, _ = h.run()
            assert order.tolist() == list(range(n + 1))


def test_run_matches_stepping(emu_api, oracle):
    emu_api.set_order_mode(0)
    n = 150
    D = oracle.synth(n, 5)
    o_ref, ev_ref, se = oracle.run(D)
    with Handle(emu_api, n, record_events=True) as h:
        h.set_matrix(D, chunk_rows=17)
        order, st = h.run()
        ev = h.events()
    assert (order == o_ref).all()
    assert st.sum_entries == se and st.n_events == len(ev_ref)
    for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id", "entries"):
        assert (ev[f] == ev_ref[f]).all(), f
    assert (ev["best"].view(np.int64) == ev_ref["best"].view(np.int64)).all()


def test_packed_upper_upload(emu_api, oracle):
    """N3: DistancesAndNames' packed strict upper triangle (DistancesAndNames.java:24-38) uploaded
    as such gives the matrix FastNN.java:307-312 builds, and the same order."""
    emu_api.set_order_mode(0)
    for n in (1, 2, 3, 4, 5, 37, 150):
        D = oracle.synth(n, 3) if n > 1 else np.zeros((1, 1))
        with Handle(emu_api, n) as h:
            h.set_packed_upper(D[np.triu_indices(n, 1)])
            if n > 3:
                h.begin()
                assert (h.live_matrix().view(np.int64) == D.view(np.int64)).all()
                while h.step() is not None:
                    pass
                order = h.finish()
            else:
                order, _ = h.run()
        assert (order == oracle.run(D)[0]).all() if n > 3 else order.tolist() == list(range(n + 1))


def test_validate_rejects_bad_matrix(emu_api, oracle):
    from fastneighbornet_amd._capi import FnnError
    D = oracle.synth(8, 1)
    for bad in ("asym", "diag", "nan"):
        E = D.copy()
        if bad == "asym":
            E[2, 5] += 1e-9
        elif bad == "diag":
            E[3, 3] = 0.5
        else:
            E[1, 4] = E[4, 1] = np.nan
        with Handle(emu_api, 8, validate=True) as h:
            h.set_matrix(E)
            with pytest.raises(FnnError):
                h.run()


@pytest.mark.parametrize("dist", ["uniform53", "dec4"])
def test_exact_rx_path_gives_the_same_trajectory(emu_api, oracle, dist):
    """The 4-candidate choice is normally certified from tree sums; force_exact_rx makes every
    event take the exact sequential-sum path instead.  Both must match the oracle."""
    emu_api.set_order_mode(2)
    for n in (9, 33, 90):
        compare_trajectory(emu_api, oracle, oracle.synth(n, 4, dist), deep=True, force_exact_rx=True)


def test_fp32_screening_keeps_the_exact_result(emu_api, oracle):
    """Events first stream the fp32 copy; only units within the error bound of the estimate's
    minimum are rescanned in fp64.  Trajectories must still match the oracle bit for bit, also
    when the candidate list overflows (tiny capacity) and for inputs whose fp32 roundings tie
    (dec4, two-valued) or are negative / large."""
    import ctypes as C
    lib = emu_api.lib
    lib.emu_set_screen_debug.argtypes = [C.c_int32, C.c_int32]
    try:
        for cap in (0, 1, 3):
            lib.emu_set_screen_debug(0, cap)
            for mode in (0, 5):
                emu_api.set_order_mode(mode)
                for n, seed, dist in [(40, 1, "uniform53"), (90, 2, "dec4"), (130, 3, "uniform53")]:
                    compare_trajectory(emu_api, oracle, oracle.synth(n, seed, dist), deep=True, deep_every=5)
        lib.emu_set_screen_debug(0, 0)
        emu_api.set_order_mode(2)
        rng = np.random.default_rng(3)
        n = 70
        A = rng.integers(1, 3, size=(n, n)).astype(np.float64); A = np.triu(A, 1); A = A + A.T
        compare_trajectory(emu_api, oracle, A, deep=True, deep_every=3)
        B = oracle.synth(n, 4) * 1e30                      # large magnitudes: bound scales with Dmax
        compare_trajectory(emu_api, oracle, B, deep=True, deep_every=3)
        Cm = oracle.synth(n, 5) - 0.5                       # negative "distances"
        np.fill_diagonal(Cm, 0.0)
        compare_trajectory(emu_api, oracle, Cm, deep=True, deep_every=3)
        Dh = oracle.synth(n, 6) * 1e300                     # beyond float range: screening must switch itself off
        compare_trajectory(emu_api, oracle, Dh, deep=True, deep_every=3)
        # statistics: screening really ran and rescanned only a fraction of the units
        with Handle(emu_api, 400, lookahead=-1) as h:
            h.set_matrix(oracle.synth(400, 7))
            order, st = h.run()
        assert st.n_screen_events > 300 and 0 < st.n_rescan_units and st.n_window_hits == 0
        with Handle(emu_api, 400, disable_screen=True) as h:
            h.set_matrix(oracle.synth(400, 7))
            order2, st2 = h.run()
        assert st2.n_screen_events == 0 and (order == order2).all()
    finally:
        lib.emu_set_screen_debug(0, 0)


def test_lookahead_windows_keep_the_exact_result(emu_api, oracle, monkeypatch):
    """A base scan also emits the pairs whose lower bound for the next K events lies under a
    threshold; the following events take their minimum from those pairs plus the rows of the
    clusters created since, and rescan when that cannot be certified.  Whatever K, the wanted list
    size and the capacity are, every trajectory must match the oracle bit for bit."""
    rng = np.random.default_rng(11)
    two = rng.integers(1, 3, size=(70, 70)).astype(np.float64); two = np.triu(two, 1); two = two + two.T
    neg = oracle.synth(60, 5) - 0.5
    np.fill_diagonal(neg, 0.0)
    cases = [(oracle.synth(n, seed, dist), every) for n, seed, dist, every in
             [(9, 1, "uniform53", 1), (33, 2, "dec4", 1), (64, 3, "uniform53", 1), (131, 4, "uniform53", 3),
              (200, 5, "dec4", 7), (330, 6, "uniform53", 17)]] + [(two, 3), (neg, 3), (np.ones((40, 40)) - np.eye(40), 1), (tree_metric(90, 4), 3)]
    total_hits = total_fails = total_stalled = 0
    for K, target, pcap, mode in [(1, 8192, 0, 0), (3, 4, 0, 2), (8, 64, 0, 4), (64, 8192, 0, 8), (64, 1, 0, 1),
                                  (512, 8192, 0, 5), (16, 8192, 7, 2), (64, 65536, 300, 6)]:
        if pcap:
            monkeypatch.setenv("FNN_LA_PCAP", str(pcap))
        else:
            monkeypatch.delenv("FNN_LA_PCAP", raising=False)
        emu_api.set_order_mode(mode)
        for D, every in cases:
            compare_trajectory(emu_api, oracle, D, deep=True, deep_every=every, lookahead=K, lookahead_pairs=target)
        n = 260
        D = oracle.synth(n, 9)
        o_ref, ev_ref, se = oracle.run(D)
        with Handle(emu_api, n, record_events=True, lookahead=K, lookahead_pairs=target) as h:
            h.set_matrix(D)
            order, st = h.run()
            ev = h.events()
        assert (order == o_ref).all() and st.sum_entries == se
        assert (ev["best"].view(np.int64) == ev_ref["best"].view(np.int64)).all()
        # every event is served by a window or scans (the last few, below 8 live nodes, scan unscreened)
        assert st.n_events - 8 <= st.n_base_scans + st.n_window_hits <= st.n_events
        total_hits += st.n_window_hits
        total_fails += st.n_window_fails
        total_stalled += st.n_stalled_events
    # windows were used, some could not certify their event, and launch sequences without scan kernels
    # that found their window gone stalled until the host relaunched with a scan
    assert total_hits > 500 and total_fails > 0 and total_stalled > 0
    # negative entries: the monotonicity argument does not hold, windows stay closed
    with Handle(emu_api, 60) as h:
        h.set_matrix(neg)
        _, st = h.run()
    assert st.n_window_hits == 0


def test_get_matrix_returns_the_resident_matrix_until_the_run_consumes_it(emu_api, oracle):
    """fnn_get_matrix (include/fastnn.h): bit-exact copy of what was uploaded; FNN_ESTATE without a resident matrix."""
    from fastneighbornet_amd._capi import FnnError
    n = 37
    D = oracle.synth(n, 2)
    with Handle(emu_api, n) as h:
        with pytest.raises(FnnError):
            h.matrix()
        h.set_matrix(D)
        assert (h.matrix().view(np.int64) == D.view(np.int64)).all()
        h.run()
        with pytest.raises(FnnError) as ei:
            h.matrix()
        assert ei.value.code == -5
