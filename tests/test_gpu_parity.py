"""GPU parity tests proper: libfastnn_hip.so through its C ABI against the oracle on the
same seeded inputs (bit-exact: event records, Sx bits, live matrix bits, final order)."""
import ctypes as C

import numpy as np
import pytest

from common import bits, check_order, compare_trajectory
from fastneighbornet_amd._capi import FnnError, Handle

pytestmark = pytest.mark.gpu


def test_device_present(hip_api):
    assert hip_api.device_count() >= 1


@pytest.mark.parametrize("dist", ["uniform53", "dec4"])
def test_small_sizes_deep(hip_api, oracle, dist):
    for n in [4, 5, 6, 7, 8, 9, 10, 13, 16, 17, 31, 32, 33, 63, 64, 65]:
        for seed in (1, 2):
            compare_trajectory(hip_api, oracle, oracle.synth(n, seed, dist), deep=True)


@pytest.mark.parametrize("n,seed,dist", [(257, 1, "uniform53"), (513, 2, "dec4"), (700, 3, "uniform53"),
                                         (1030, 4, "dec4")])
def test_medium_trajectory(hip_api, oracle, n, seed, dist):
    # crosses the 512-column / 32-row scan tile boundaries and the 1024 threshold of
    # NetMakerOriginal.java:361
    k, order = compare_trajectory(hip_api, oracle, oracle.synth(n, seed, dist), deep=True, deep_every=101)
    check_order(order, n)


@pytest.mark.parametrize("n,seed", [(1500, 1), (2048, 7)])
def test_run_matches_oracle(hip_api, oracle, n, seed):
    D = oracle.synth(n, seed)
    o_ref, ev_ref, se = oracle.run(D, threads=8)
    with Handle(hip_api, n, validate=True, record_events=True) as h:
        h.set_matrix(D, chunk_rows=333)
        order, st = h.run()
        ev = h.events()
    assert (order == o_ref).all()
    assert st.sum_entries == se and st.n_events == len(ev_ref)
    for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id", "entries"):
        assert (ev[f] == ev_ref[f]).all(), f
    assert (bits(ev["best"]) == bits(ev_ref["best"])).all()


@pytest.mark.parametrize("dist", ["uniform53", "dec4"])
def test_exact_rx_path(hip_api, oracle, dist):
    # force the rare path (exact ComputeRx sums) on every event
    for n in (9, 65, 300):
        compare_trajectory(hip_api, oracle, oracle.synth(n, 4, dist), deep=True, deep_every=7, force_exact_rx=True)
    compare_trajectory(hip_api, oracle, oracle.synth(1200, 5, dist), deep=False, force_exact_rx=True)


def test_device_synth_is_bit_identical(hip_api, oracle):
    for n, seed, dist in [(100, 1, "uniform53"), (777, 9, "dec4")]:
        D = oracle.synth(n, seed, dist)
        o_ref, _, _ = oracle.run(D, threads=8)
        with Handle(hip_api, n) as h:
            h.synth(seed, dist)
            assert (bits(h.matrix()) == bits(D)).all()   # fnn_get_matrix: the resident matrix, before the run consumes it
            h.begin()
            assert (bits(h.live_matrix()) == bits(D)).all()
            while h.step() is not None:
                pass
            assert (h.finish() == o_ref).all()


def test_packed_upper_upload(hip_api, oracle):
    """N3: the packed strict upper triangle of DistancesAndNames (DistancesAndNames.java:24-38) is
    mirrored on the device; n = 6000 needs two staged chunks."""
    for n in (2, 4, 5, 100, 777, 6000):
        D = oracle.synth(n, 4)
        with Handle(hip_api, n) as h:
            h.set_packed_upper(D[np.triu_indices(n, 1)])
            if n < 4:
                assert h.run()[0].tolist() == list(range(n + 1))
                continue
            h.begin()
            assert (bits(h.live_matrix()) == bits(D)).all()
            if n <= 777:
                while h.step() is not None:
                    pass
                assert (h.finish() == oracle.run(D, threads=8)[0]).all()


def test_one_call_and_host_mirror(hip_api, oracle):
    import fastneighbornet_amd as fa
    n = 300
    D = oracle.synth(n, 11)
    o_ref, _, _ = oracle.run(D)
    D0 = D.copy()
    assert (fa.canonical_order(D) == o_ref).all()
    nn = fa.NeighborNetCanonical(D, n, 1, None)
    assert (nn.runNeighborNet() == o_ref).all()
    assert (D == D0).all()  # the caller's matrix is not consumed
    for k in (1, 2, 3):
        assert fa.NeighborNetCanonical(D[:k, :k], k).runNeighborNet().tolist() == list(range(k + 1))


def test_validate_rejects_bad_matrix(hip_api, oracle):
    D = oracle.synth(40, 1)
    for bad in ("asym", "diag", "inf"):
        E = D.copy()
        if bad == "asym":
            E[2, 35] += 1e-9
        elif bad == "diag":
            E[33, 33] = 0.5
        else:
            E[1, 4] = E[4, 1] = np.inf
        with Handle(hip_api, 40, validate=True) as h:
            h.set_matrix(E)
            with pytest.raises(FnnError):
                h.run()


def test_call_sequence_errors(hip_api):
    with Handle(hip_api, 10) as h:
        with pytest.raises(FnnError):
            h.run()  # no matrix
    with pytest.raises(FnnError):
        Handle(hip_api, -1)


def test_tie_rich_input(hip_api, oracle):
    # all distances equal: every Q ties, the result is decided purely by scan order
    n = 67
    D = np.full((n, n), 0.25)
    np.fill_diagonal(D, 0.0)
    compare_trajectory(hip_api, oracle, D, deep=True)
    # two-valued matrix
    rng = np.random.default_rng(5)
    A = rng.integers(1, 3, size=(n, n)).astype(np.float64)
    A = np.triu(A, 1)
    A = A + A.T
    compare_trajectory(hip_api, oracle, A, deep=True)


def test_full_size_invariants(hip_api):
    """BASELINE.json size (32768 taxa, 8 GiB fp64): size-independent properties."""
    n = 32768
    with Handle(hip_api, n, record_events=True) as h:
        h.synth(1, "uniform53")
        order, st = h.run()
        ev = h.events()
    check_order(order, n)
    assert st.n_window_hits > 20000 and st.n_base_scans < 5000
    # lookahead windows, per-event screening + exact rescans, and the plain fp64 scan of every
    # event must give the same trajectory event by event
    for kw in (dict(lookahead=-1), dict(disable_screen=True)):
        with Handle(hip_api, n, record_events=True, **kw) as h:
            h.synth(1, "uniform53")
            order_plain, st_plain = h.run()
            ev_plain = h.events()
        assert st_plain.n_window_hits == 0
        assert (st_plain.n_screen_events == 0) if "disable_screen" in kw else (st_plain.n_screen_events > 20000)
        assert (order == order_plain).all()
        for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id"):
            assert (ev[f] == ev_plain[f]).all(), f
        assert (bits(ev["best"]) == bits(ev_plain["best"])).all()
    assert st.n_events == len(ev)
    # clusters drop by exactly one per event; c <= m <= 2c; counters follow the event kinds
    assert (np.diff(ev["c_before"]) == -1).all()
    assert (ev["c_before"] <= ev["m_before"]).all() and (ev["m_before"] <= 2 * ev["c_before"]).all()
    dm = {2: 0, 3: -1, 4: -2}
    for k in (2, 3, 4):
        sel = np.nonzero(ev["kind"][:-1] == k)[0]
        assert (ev["m_before"][sel + 1] - ev["m_before"][sel] == dm[k]).all()
    m, c = ev["m_before"].astype(np.int64), ev["c_before"].astype(np.int64)
    e = m * (m - 1) // 2 - (m - c)
    e[ev["kind"] == 5] = 0
    assert (ev["entries"] == e).all() and st.sum_entries == e.sum()
    assert n ** 3 / 6 <= st.sum_entries <= n ** 3 / 3


def test_fp32_screening_small_sizes(oracle):
    """Screening forced on for small matrices (separate process: the thresholds are read from
    the environment when the handle is created)."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.environ["FNN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FNN_ROOT"], "tests"))
import numpy as np
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
from common import compare_trajectory, tree_metric
a = fa.api()
for n, seed, dist in [(40, 1, "uniform53"), (300, 2, "dec4"), (1100, 3, "uniform53"), (2100, 4, "uniform53")]:
    compare_trajectory(a, O, O.synth(n, seed, dist), deep=(n < 1000), deep_every=11)
# lookahead windows: every combination of window length / wanted list size / capacity must give the
# oracle's trajectory (k_track, the emission pass of k_screen, window failures, list overflow)
for K, target, pcap in [(-1, 0, 0), (1, 0, 0), (3, 4, 0), (8, 64, 0), (64, 1, 0), (512, 60000, 0), (16, 8192, 7), (64, 65536, 300)]:
    if pcap: os.environ["FNN_LA_PCAP"] = str(pcap)
    else: os.environ.pop("FNN_LA_PCAP", None)
    for n, seed, dist in [(33, 1, "uniform53"), (200, 2, "dec4"), (700, 3, "uniform53")]:
        compare_trajectory(a, O, O.synth(n, seed, dist), deep=(n < 300), deep_every=7, lookahead=K, lookahead_pairs=target)
os.environ.pop("FNN_LA_PCAP", None)
# whole runs (no per-event host round trip): batches of launch sequences, the new cluster's exact row
# sum computed beside the next event's tracking, stalls and recoveries as in production
def whole_run(n, seed, dist, **kw):
    D = O.synth(n, seed, dist)
    o_ref, ev_ref, se = O.run(D, threads=4)
    with Handle(a, n, record_events=True, **kw) as h:
        h.set_matrix(D)
        order, st = h.run()
        ev = h.events()
    assert (order == o_ref).all(), (n, seed, dist, kw)
    assert st.n_events == len(ev_ref) and st.sum_entries == se
    for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id", "entries"):
        assert (ev[f] == ev_ref[f]).all(), (f, n, seed, dist, kw)
    assert (ev["best"].view(np.int64) == ev_ref["best"].view(np.int64)).all(), (n, seed, dist, kw)
    return st
for K, target in [(0, 0), (5, 16), (64, 60000), (300, 0)]:
    for n, seed, dist in [(33, 1, "uniform53"), (130, 2, "dec4"), (700, 3, "uniform53"), (1500, 4, "uniform53")]:
        whole_run(n, seed, dist, lookahead=K, lookahead_pairs=target)
st = whole_run(2500, 5, "uniform53")
assert st.n_window_hits > 1500
rng = np.random.default_rng(5)
A = rng.integers(1, 3, size=(400, 400)).astype(np.float64); A = np.triu(A, 1); A = A + A.T
with Handle(a, 400) as h:
    h.set_matrix(A); o1, _ = h.run()
assert (o1 == O.run(A)[0]).all()
# additive tree metrics with dyadic branch lengths: path sums are exact, so the Q criterion has
# exact ties between cherries - windows, certified decisions and tie-breaks under stress
for n, seed in [(60, 1), (257, 2), (600, 3)]:
    compare_trajectory(a, O, tree_metric(n, seed), deep=(n < 100), deep_every=5)
rng = np.random.default_rng(3)
n = 300
A = rng.integers(1, 3, size=(n, n)).astype(np.float64); A = np.triu(A, 1); A = A + A.T
compare_trajectory(a, O, A, deep=False)
compare_trajectory(a, O, O.synth(n, 4) * 1e30, deep=False)
Cm = O.synth(n, 5) - 0.5; np.fill_diagonal(Cm, 0.0)
compare_trajectory(a, O, Cm, deep=False)
compare_trajectory(a, O, O.synth(n, 6) * 1e300, deep=False)
with Handle(a, 3000) as h:
    h.synth(7, "uniform53"); order, st = h.run()
o_ref, _, _ = O.run(O.synth(3000, 7), threads=8)
assert (order == o_ref).all()
assert st.n_window_hits > 2000 and st.n_base_scans > 20 and st.n_rescan_units > 0
with Handle(a, 3000, lookahead=-1) as h:
    h.synth(7, "uniform53"); order, st = h.run()
assert (order == o_ref).all()
assert st.n_screen_events > 2000 and st.n_window_hits == 0
print("SCREEN_OK", st.n_screen_events, st.n_rescan_units)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FNN_ROOT=root, FNN_SCREEN_MIN_N="8", FNN_SCREEN_MIN_M="8")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SCREEN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_screened_run_equals_plain_fp64_run(hip_api):
    """Above 8192 live nodes the scan goes through the screening pass + exact rescans.  The
    whole trajectory must be identical to a run that scans the fp64 matrix in full."""
    n = 10240
    runs = []
    for disable in (True, False):
        with Handle(hip_api, n, record_events=True, disable_screen=disable) as h:
            h.synth(3, "uniform53")
            order, st = h.run()
            runs.append((order, h.events(), st))
    (o0, e0, s0), (o1, e1, s1) = runs
    assert s0.n_screen_events == 0 and s0.n_window_hits == 0
    assert s1.n_window_hits > 1000 and s1.n_base_scans > 10   # lookahead windows served most screened events
    assert (o0 == o1).all()
    for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id", "entries"):
        assert (e0[f] == e1[f]).all(), f
    assert (bits(e0["best"]) == bits(e1["best"])).all()


def test_screened_run_against_the_oracle(hip_api, oracle):
    """n just above the screening threshold: the first ~45 % of the events go through the bf16
    screening pass + exact rescans; the whole trajectory must equal the oracle's."""
    n = 8704
    D = oracle.synth(n, 12)
    o_ref, ev_ref, se = oracle.run(D, threads=16)
    with Handle(hip_api, n, validate=True, record_events=True) as h:
        h.set_matrix(D, chunk_rows=1024)
        order, st = h.run()
        ev = h.events()
    assert st.n_window_hits > 1000
    assert (order == o_ref).all()
    assert st.sum_entries == se and st.n_events == len(ev_ref)
    for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id", "entries"):
        assert (ev[f] == ev_ref[f]).all(), f
    assert (bits(ev["best"]) == bits(ev_ref["best"])).all()


@pytest.mark.parametrize("n,seed,dist", [(8448, 22, "dec4")])
def test_window_runs_against_the_oracle(hip_api, oracle, n, seed, dist):
    """Lookahead windows at sizes where they serve most events (and, with the 4-decimal input, under
    exact ties of Q): the whole trajectory must equal the oracle's."""
    D = oracle.synth(n, seed, dist)
    o_ref, ev_ref, se = oracle.run(D, threads=16)
    with Handle(hip_api, n, record_events=True) as h:
        h.set_matrix(D, chunk_rows=2048)
        order, st = h.run()
        ev = h.events()
    assert (order == o_ref).all()
    assert st.sum_entries == se and st.n_events == len(ev_ref)
    for f in ("m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id", "entries"):
        assert (ev[f] == ev_ref[f]).all(), f
    assert (bits(ev["best"]) == bits(ev_ref["best"])).all()
    assert st.n_window_hits + st.n_base_scans >= st.n_events - 4096 - 8
