"""Input classes beyond the bench's uniform matrices, shared by tests/golden/make_golden_big.py (which froze the
oracle's results for them in the build container) and the GPU tests that replay them.

  uniform53, dec4   SURVEY.md 8(d) generators; the engine generates them on the device from the seed
  tree              additive tree metric, dyadic branch lengths: path sums are exact, so the Q criterion has exact
                    ties everywhere (NeighborNetCanonical.java:151-178 first-strict-minimum, NetMakerOriginal.java:428-452)
  treenoise         additive tree metric with real-valued branch lengths + 5 % uniform noise (what real data look like)
  neg               uniform53 shifted by -0.25: a third of the entries negative (the lookahead windows' monotonicity
                    argument needs non-negative entries: the engine takes the plain fp64 scan for every event - no screening
                    pass, no windows; with several ranks the scan of every event is sharded -)

Host-generated classes depend on numpy's default_rng stream (PCG64), which is stable across numpy versions.
"""
import hashlib

import numpy as np

DEVICE_DISTS = ("uniform53", "dec4")


def _tree(n, seed, dyadic):
    r = np.random.default_rng(seed)
    D = np.zeros((n, n))
    depth = np.zeros(n)
    stack = [(0, n, 0.0)]
    while stack:
        a, b, h = stack.pop()           # h = length of the path from the root to this node
        if b - a == 1:
            depth[a] = h
            continue
        c = int(r.integers(a + 1, b))
        la = float(r.integers(1, 64)) / 64.0 if dyadic else float(r.random()) + 0.01
        lb = float(r.integers(1, 64)) / 64.0 if dyadic else float(r.random()) + 0.01
        stack.append((a, c, h + la))
        stack.append((c, b, h + lb))
        D[a:c, c:b] -= 2.0 * h          # -2 h(lca); the leaf depths are added below
    iu = np.triu_indices(n, 1)
    D[iu] += depth[iu[0]] + depth[iu[1]]
    D = np.triu(D, 1)
    D = D + D.T
    p = r.permutation(n)
    return np.ascontiguousarray(D[np.ix_(p, p)])


def make(n, dist, seed, oracle):
    """The n x n matrix of an input class (host side; `oracle` = the oracle module, for the SplitMix64 generators)."""
    if dist in DEVICE_DISTS:
        return oracle.synth(n, seed, dist)
    if dist == "tree":
        return _tree(n, seed, True)
    if dist == "treenoise":
        T = _tree(n, seed, False)
        N = np.triu(np.random.default_rng(seed + 1000).random((n, n)) * 0.05, 1)
        return np.ascontiguousarray(T + N + N.T)
    if dist == "neg":
        D = oracle.synth(n, seed, "uniform53")
        D -= 0.25
        np.fill_diagonal(D, 0.0)
        return D
    raise ValueError(dist)


def sha_big(a):
    h = hashlib.sha256()
    flat = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    step = 1 << 28
    for o in range(0, flat.size, step):
        h.update(flat[o:o + step].tobytes())
    return h.hexdigest()
