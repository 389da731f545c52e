"""Relaxed mode: `-mode Relaxed` without `-additive`, NeighborNetLocal.java:88-264.

The reference draws from ThreadLocalRandom (unseedable): no two runs of the reference agree, so there is no reference
output to compare with even in principle.  Oracle and engine both draw from java.util.Random(seed) instead (the
generator of the line the reference commented out, NeighborNetLocal.java:27); given the seed the whole run is
deterministic and the engine must reproduce the oracle's trajectory bit for bit.  The generator itself is pinned by
java.util.Random's well-known outputs.
"""
import numpy as np
import pytest

from common import check_order
from fastneighbornet_amd._capi import Handle

TRAJ = ["m_before", "c_before", "cx_id", "cy_id", "x_id", "y_id", "kind", "u_id"]


def test_java_util_random_known_answers(oracle):
    # new Random(42).nextInt(10) x 10 and new Random(0).nextInt(100) x 8: the sequences every Java tutorial prints
    assert oracle.java_random_ints(42, [10] * 10) == [0, 3, 8, 4, 0, 5, 5, 8, 9, 3]
    assert oracle.java_random_ints(0, [100] * 8) == [60, 48, 29, 47, 15, 53, 91, 61]
    # power-of-two bounds take the multiply-shift branch
    assert oracle.java_random_ints(42, [16] * 3) == [11, 0, 10]


def circ_instance(n, seed):
    """A circular (Kalmanson) metric: split weights on a random circular order -> every mode must recover the order."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n)
    w = rng.random((n, n)) + 0.05
    D = np.zeros((n, n))
    # split {perm[i..j-1]} | rest for 0 <= i < j <= n-1 (positions on the circle)
    c = np.zeros((n + 1, n + 1))
    for i in range(n):
        for j in range(i + 1, n):
            c[i, j] = w[i, j]
    # distance between circle positions a < b: sum of splits [i, j) with i <= a < j <= b ... done densely for small n
    for a in range(n):
        for b in range(a + 1, n):
            s = 0.0
            for i in range(0, a + 1):
                for j in range(a + 1, b + 1):
                    s += c[i, j]
            for i in range(a + 1, b + 1):
                for j in range(b + 1, n):
                    s += c[i, j]
            D[perm[a], perm[b]] = D[perm[b], perm[a]] = s
    return D, perm


def same_circular_order(order, perm):
    n = len(perm)
    got = [int(t) - 1 for t in order[1:]]
    want = [int(p) for p in perm]
    i = want.index(got[0])
    fwd = want[i:] + want[:i]
    bwd = [want[(i - k) % n] for k in range(n)]
    return got == fwd or got == bwd


def test_oracle_relaxed_runs_and_recovers_circular_orders(oracle):
    for n, seed in [(8, 1), (40, 2), (200, 3)]:
        D = oracle.synth(n, seed)
        order, ev = oracle.run_relaxed(D, 7, 8)
        check_order(order, n)
        o2, ev2 = oracle.run_relaxed(D, 7, 8)
        assert (order == o2).all() and ev.tobytes() == ev2.tobytes()       # deterministic given the seed
        o3, _ = oracle.run_relaxed(D, 8, 8)
        check_order(o3, n)
    # below the threshold the Relaxed mode IS the Canonical one (NetMakerOriginal.java:361)
    D = oracle.synth(300, 5)
    assert (oracle.run_relaxed(D, 1, 0)[0] == oracle.run(D)[0]).all()
    # a circular metric: the relaxed search merges mutual row minima only, which Neighbor-Net's consistency covers
    D, perm = circ_instance(24, 4)
    for seed in (1, 2, 3):
        order, _ = oracle.run_relaxed(D, seed, 6)
        assert same_circular_order(order, perm)


def compare(api, oracle, D, seed, min_active, **kw):
    n = D.shape[0]
    o_ref, ev_ref = oracle.run_relaxed(D, seed, min_active)
    with Handle(api, n, record_events=True, relaxed_seed=seed, relaxed_min_active=min_active, **kw) as h:
        h.set_matrix(D)
        order, st = h.run()
        ev = h.events()
    assert len(ev) == len(ev_ref)
    for f in TRAJ + ["entries"]:
        bad = np.nonzero(ev[f] != ev_ref[f])[0]
        assert bad.size == 0, f"{f} differs first at event {bad[0]}: {ev[f][bad[0]]} vs {ev_ref[f][bad[0]]}"
    assert (np.ascontiguousarray(ev["best"]).view(np.int64) == np.ascontiguousarray(ev_ref["best"]).view(np.int64)).all()
    assert (order == o_ref).all()
    return st


@pytest.mark.parametrize("n,seed,dist,min_active", [(5, 1, "uniform53", 3), (9, 2, "uniform53", 4), (33, 3, "dec4", 4),
                                                    (120, 4, "uniform53", 8), (257, 5, "dec4", 16), (600, 6, "uniform53", 64)])
def test_emulation_matches_oracle(emu_api, oracle, n, seed, dist, min_active):
    compare(emu_api, oracle, oracle.synth(n, seed, dist), 1000 + seed, min_active)


@pytest.mark.parametrize("n,dist", [(40, "uniform53"), (150, "dec4")])
def test_emulation_deep_state_matches_oracle(emu_api, oracle, n, dist):
    """after every event: node ids, partners, Sx bits and the live matrix bits (the merges are the Canonical ones, but
    here Cx / Cy need not be their clusters' representatives)"""
    from common import compare_trajectory
    compare_trajectory(emu_api, oracle, oracle.synth(n, 11, dist), relaxed_seed=77, relaxed_min_active=4)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dist", [(65, "uniform53"), (300, "dec4")])
def test_gpu_deep_state_matches_oracle(hip_api, oracle, n, dist):
    from common import compare_trajectory
    compare_trajectory(hip_api, oracle, oracle.synth(n, 12, dist), relaxed_seed=78, relaxed_min_active=4)


def test_emulation_default_threshold(emu_api, oracle):
    compare(emu_api, oracle, oracle.synth(1300, 9), 5, 0)   # 1300 -> 1024 relaxed, then the full scans


def ev_m(api, oracle, n, seed, dist, min_active):
    _, ev = oracle.run_relaxed(oracle.synth(n, seed, dist), 2000 + seed, min_active)
    return ev["m_before"][ev["kind"] != 5]


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,dist,min_active", [(5, 1, "uniform53", 3), (33, 3, "dec4", 4), (257, 5, "dec4", 16),
                                                    (600, 6, "uniform53", 64), (1500, 7, "uniform53", 0),
                                                    (3000, 8, "dec4", 0), (5000, 9, "uniform53", 1024)])
def test_gpu_matches_oracle(hip_api, oracle, n, seed, dist, min_active):
    st = compare(hip_api, oracle, oracle.synth(n, seed, dist), 2000 + seed, min_active)
    assert st.n_window_hits == 0
    thr = min_active if min_active > 0 else 1024
    assert st.n_relaxed_events == int((ev_m(hip_api, oracle, n, seed, dist, min_active) > thr).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [2, 3, 16])
@pytest.mark.parametrize("n,seed,dist,min_active", [(257, 5, "dec4", 16), (2100, 6, "uniform53", 64), (3000, 8, "dec4", 0)])
def test_gpu_row_pass_on_several_workgroups(hip_api, oracle, monkeypatch, grid, n, seed, dist, min_active):
    """the row pass of a minimum spread over `grid` workgroups (the default from ~12 000 live nodes on): command word,
    per-workgroup records, merged tie lists - same trajectory"""
    monkeypatch.setenv("FNN_RELAXED_GRID", str(grid))
    compare(hip_api, oracle, oracle.synth(n, seed, dist), 3000 + seed, min_active)


@pytest.mark.gpu
def test_gpu_matches_oracle_at_13000_taxa_with_the_shipped_grid_rule(hip_api, oracle):
    """from 12 288 live nodes on the row pass runs on m / 4096 workgroups by default: the first ~700 events of this run"""
    compare(hip_api, oracle, oracle.synth(13000, 4), 9, 0)


@pytest.mark.gpu
def test_gpu_relaxed_is_canonical_below_the_threshold(hip_api, oracle):
    D = oracle.synth(700, 2)
    with Handle(hip_api, 700, relaxed_seed=3) as h:
        h.set_matrix(D)
        order, _ = h.run()
    assert (order == oracle.run(D)[0]).all()


@pytest.mark.gpu
def test_gpu_cli_and_host_mirrors_in_relaxed_mode(oracle, tmp_path):
    """`fastnn -mode Relaxed -seed S -order` and NeighborNetLocal(...).runNeighborNet() (FastNN.java:329-338)."""
    import os
    import subprocess

    import fastneighbornet_amd as fa
    from test_host_cli import PKG, write_phylip
    n, seed = 1100, 5
    D = oracle.synth(n, 3)
    p = str(tmp_path / "r.phy")
    write_phylip(p, D)
    D2 = D  # (write_phylip prints repr(): the reader gets the same doubles back)
    o_ref, _ = oracle.run_relaxed(D2, seed, 0)
    r = subprocess.run([os.path.join(PKG, "bin", "fastnn"), "-distFile", p, "-mode", "Relaxed", "-seed", str(seed), "-order", "-time"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "[" + ", ".join(str(int(v)) for v in o_ref) + "]\n"
    assert "Using the relaxed version without additivity checking.\n" in r.stderr
    nn = fa.NeighborNetLocal(D2, n, 1, False, None, seed=seed)
    assert (nn.runNeighborNet() == o_ref).all() and nn.stats["n_relaxed_events"] > 0
    with pytest.raises(NotImplementedError):
        fa.NeighborNetLocal(D2, n, 1, True, None, seed=seed)


def test_relaxed_mode_rejects_several_ranks_and_bad_modes(emu_api):
    import ctypes as C

    from fastneighbornet_amd._capi import FnnError, FnnOpts
    with Handle(emu_api, 64, relaxed_seed=1) as h:
        with pytest.raises(FnnError, match="one GPU"):
            h.comm_init_host(2, 0, lambda send: [send, send])
        h.comm_init_host(1, 0, lambda send: [send])   # a world of one is fine
    opts = FnnOpts()
    opts.mode = 7
    out = C.c_void_p()
    assert emu_api.create(8, C.byref(opts), C.byref(out)) < 0
