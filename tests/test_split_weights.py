"""Circular split weights.  CPU part: the oracle (oracle/csw_oracle.c, a
restatement of CircularSplitWeights.java with the distance re-ordering restored) is pinned
mathematically - against a dense non-negative least-squares solve of the reference's LIVE design
matrix (FastNN.java:405-437; the optimum is unique) and against the known weights of synthetic
circular metrics.  GPU part: fnn_split_weights_f64 against the oracle."""
import numpy as np
import pytest

from oracle import csw_oracle as W


def circ_instance(n, seed, density=0.4):
    """Random hidden circular order + random non-negative split weights -> (D, ordering, weights)."""
    rng = np.random.default_rng(seed)
    order = np.concatenate([[0, 1], 2 + rng.permutation(n - 1)]).astype(np.int32)
    A = W.live_design_matrix(n, order)
    w = rng.random(A.shape[1]) * (rng.random(A.shape[1]) < density)
    b = A @ w
    D = np.zeros((n, n))
    k = 0
    for a in range(n):
        for c in range(a + 1, n):
            D[a, c] = D[c, a] = b[k]
            k += 1
    return D, order, w


def test_operators_are_adjoint_and_match_the_dense_matrix():
    rng = np.random.default_rng(1)
    for n in (4, 5, 8, 13):
        ident = np.arange(n + 1, dtype=np.int32)
        A_live = W.live_design_matrix(n, ident)
        # live column k=(i,j) <-> fast split: (i-1,j-1) for i>=1, (j-1,n-1) for i=0
        cols = []
        for i in range(n):
            for j in range(i + 1, n):
                fi, fj = (i - 1, j - 1) if i >= 1 else (j - 1, n - 1)
                cols.append((2 * n - fi - 3) * fi // 2 + fj - 1)
        A_fast = np.zeros_like(A_live)
        A_fast[:, cols] = A_live
        x = rng.random(W.npairs(n))
        y = rng.random(W.npairs(n))
        assert np.allclose(W.calculate_ab(n, x), A_fast @ x, rtol=1e-12, atol=1e-12)
        assert np.allclose(W.calculate_atx(n, y), A_fast.T @ y, rtol=1e-12, atol=1e-12)
        assert abs(W.calculate_ab(n, x) @ y - x @ W.calculate_atx(n, y)) < 1e-9


def test_oracle_matches_dense_nnls_of_the_live_design_matrix(oracle):
    import scipy.optimize as so
    for n, seed in [(4, 1), (5, 2), (6, 3), (8, 4), (11, 5), (14, 6)]:
        D = oracle.synth(n, seed)
        order, _, _ = oracle.run(D)
        A = W.live_design_matrix(n, order)
        xs, _ = so.nnls(A, W.packed_distances(D), maxiter=200000)
        xl, st = W.split_weights(D, order)
        assert np.abs(xs - xl).max() < 1e-6 * max(1.0, np.abs(xs).max()), (n, np.abs(xs - xl).max())
        assert (xl >= 0).all()


def test_oracle_recovers_known_circular_weights():
    for n, seed in [(6, 1), (9, 2), (15, 3), (24, 4)]:
        D, order, w = circ_instance(n, seed)
        xl, st = W.split_weights(D, order)
        assert np.abs(xl - w).max() < 1e-9, (n, np.abs(xl - w).max())


@pytest.mark.gpu
def test_gpu_split_weights_match_the_oracle(hip_api, oracle):
    import fastneighbornet_amd as fa
    # random distances with the Canonical order of the engine itself
    import scipy.optimize as so
    # The method's own stopping rule (CG_EPSILON = 1e-8 on the residual of the normal equations) leaves
    # the weights short of the true optimum by ~1e-6 at 20 taxa, ~6e-6 at 48, ~1e-4 beyond 200 (the CPU
    # oracle shows the same distance to a dense NNLS solve), and two correct executions in different
    # summation orders differ by as much.  So: tight where the problem is well conditioned, the
    # method's accuracy elsewhere - and always the same quality of fit.
    for n, seed, tol in [(5, 1, 1e-6), (9, 2, 1e-6), (12, 7, 1e-6), (33, 3, 2e-5), (64, 4, 2e-5), (150, 5, 1e-4)]:
        D = oracle.synth(n, seed)
        order = fa.canonical_order(D)
        ref, st_ref = W.split_weights(D, order)
        got, st = fa.split_weights(D, order)
        assert np.abs(got - ref).max() < tol, (n, np.abs(got - ref).max(), st, st_ref)
        assert abs(st["nsplits"] - int((ref > 1e-6).sum())) <= 2
        if n <= 33:  # against the true optimum of the live path's dense problem
            xs, _ = so.nnls(W.live_design_matrix(n, order), W.packed_distances(D), maxiter=10 ** 7)
            assert np.abs(got - xs).max() < max(tol, 5e-6), (n, np.abs(got - xs).max())
    # Beyond ~200 taxa the normal equations are so ill-conditioned that the reference's own stopping
    # rule (CG_EPSILON = 1e-8 on the residual of A^T A x = A^T d) fixes the weights only to ~1e-4: two
    # correct executions of the same algorithm (other summation order) differ that much.  Both must
    # then be equally good solutions: same active-set path, same fit.
    n = 257
    D = oracle.synth(n, 6)
    order = fa.canonical_order(D)
    ref, st_ref = W.split_weights(D, order)
    got, st = fa.split_weights(D, order)
    assert np.abs(got - ref).max() < 2e-3
    assert st["outer_iterations"] == st_ref[0] and st["cg_calls"] == st_ref[1]
    d = W.setup_d(D, order)

    def fit(live):  # residual sum of squares of the live-order weights
        A_x = np.zeros(W.npairs(n))
        # back to the fast index space: live (i,j) -> fast (i-1,j-1) / (j-1,n-1)
        x = np.zeros(W.npairs(n))
        k = 0
        for i in range(n):
            for j in range(i + 1, n):
                fi, fj = (i - 1, j - 1) if i >= 1 else (j - 1, n - 1)
                x[(2 * n - fi - 3) * fi // 2 + fj - 1] = live[k]
                k += 1
        A_x = W.calculate_ab(n, x)
        return float(((A_x - d) ** 2).sum())
    f_ref, f_got = fit(ref), fit(got)
    assert abs(f_got - f_ref) <= 1e-7 * max(f_ref, 1e-30), (f_ref, f_got)
    # circular metrics: the known weights come back
    for n, seed in [(12, 7), (40, 8), (120, 9)]:
        D, order, w = circ_instance(n, seed)
        got, st = fa.split_weights(D, order)
        assert np.abs(got - w).max() < 1e-6 * max(1.0, w.max()), (n, np.abs(got - w).max())


@pytest.mark.gpu
def test_gpu_split_weights_argument_errors(hip_api, oracle):
    import fastneighbornet_amd as fa
    from fastneighbornet_amd._capi import FnnError
    D = oracle.synth(8, 1)
    bad = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7], dtype=np.int32)   # not a permutation
    with pytest.raises(FnnError):
        fa.split_weights(D, bad)
