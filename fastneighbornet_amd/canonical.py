"""Host-side mirror of the reference's plug-in seam for the Canonical path.

Reference: `class NeighborNetCanonical extends NetMakerOriginal`
(NeighborNetCanonical.java:29-36), constructed by FastNN.main for `-mode Canonical`
(FastNN.java:324-328) and driven by one call to `runNeighborNet()`
(NetMakerOriginal.java:129-162, FastNN.java:378/:391).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


class NeighborNetCanonical:
    """Same constructor shape and method name as the Java class.

    d          -- n x n symmetric fp64 distance matrix with zero diagonal (`double[][] d`).
                  Unlike the reference (NetMakerOriginal.java:653-656) the caller's array
                  is not modified.
    numTaxa    -- number of taxa (`ntax`)
    numThreads -- accepted for signature compatibility; the GPU engine always produces the
                  `-threads 1` result (the reference's pool branch is not deterministic,
                  SURVEY.md F7)
    pool       -- ignored (the Java ExecutorService)
    """

    def __init__(self, d, numTaxa: int, numThreads: int = 1, pool=None, *, device: int = 0,
                 validate: bool = True, record_events: bool = False):
        from . import api
        self._api = api()
        self.ntax = int(numTaxa)
        self.numThreads = int(numThreads)
        self.pool = pool
        self.D = np.ascontiguousarray(d, dtype=np.float64)
        if self.D.shape != (self.ntax, self.ntax):
            raise ValueError(f"d must be {self.ntax} x {self.ntax}")
        self._device = device
        self._validate = validate
        self._record = record_events
        self._relaxed_seed = None
        self.ordering = None
        self.stats = None
        self.events = None

    def runNeighborNet(self) -> np.ndarray:
        """int[ntax+1]: ordering[0] = 0, ordering[1] = 1, 1-based ids in circular order."""
        if self.ntax <= 3:  # NetMakerOriginal.java:133-140
            self.ordering = np.arange(self.ntax + 1, dtype=np.int32)
            return self.ordering
        with _capi.Handle(self._api, self.ntax, device=self._device, validate=self._validate,
                          record_events=self._record, relaxed_seed=self._relaxed_seed) as h:
            h.set_matrix(self.D)
            order, st = h.run()
            if self._record:
                self.events = h.events()
        self.ordering = order
        self.stats = st.as_dict()
        return order

    def getOrdering(self):  # NetMakerOriginal.java:72-74
        return self.ordering


class NeighborNetLocal(NeighborNetCanonical):
    """`-mode Relaxed` (NeighborNetLocal.java:25-32; FastNN.java:329-338): same constructor shape as the Java class
    plus `seed`.  The reference draws from ThreadLocalRandom, which cannot be seeded; the engine draws from
    java.util.Random(seed) - the generator of the line the reference commented out (:27) - so a run can be
    repeated.  `additive=True` (the additivity check) is not provided."""

    def __init__(self, d, numTaxa: int, numThreads: int = 1, additive: bool = False, pool=None, *, seed: int = 0,
                 device: int = 0, validate: bool = True, record_events: bool = False):
        if additive:
            raise NotImplementedError("the additivity check of the relaxed search (-additive) is not provided")
        super().__init__(d, numTaxa, numThreads, pool, device=device, validate=validate, record_events=record_events)
        self._relaxed_seed = int(seed)


def canonical_order(D: np.ndarray, device: int = 0, validate: bool = True) -> np.ndarray:
    """One-call form over `fnn_canonical_order_f64`."""
    from . import api
    a = api()
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    order = np.zeros(n + 1, dtype=np.int32)
    opts = _capi.FnnOpts()
    opts.device = device
    opts.validate = 1 if validate else 0
    a.check(a.canonical_order_f64(D.ctypes.data_as(C.POINTER(C.c_double)), n, n, C.byref(opts),
                                  order.ctypes.data_as(C.POINTER(C.c_int32)), None))
    return order


def split_weights(D: np.ndarray, ordering: np.ndarray, device: int = 0, allow_inexact: bool = False):
    """Non-negative least-squares weights of the circular splits of `ordering` over
    `fnn_split_weights_f64` (the optimum the reference's live path computes, FastNN.java:401-454, in its
    index order :405-419).  Returns (weights[n(n-1)/2], stats dict); stats["method"]: "closed form"
    (the unconstrained optimum is feasible), "from below" (the block active-set method on an inverse Cholesky
    factor of the free set) or "reference" (CircularSplitWeights.java's active-set /
    conjugate-gradient method); stats["refactorizations"] = rebuilds of the factor, stats["solves"] = sub-problems;
    stats["certified"] / stats["kkt_violation"]: the solver's own Kuhn-Tucker check of the returned weights.  Raises FnnError
    with code -6 (FNN_ECAPACITY: the optimum has more positive splits than the block method's factor holds and the
    reference's route is not affordable at this size) or -7 (FNN_EINEXACT, unless allow_inexact)."""
    from . import api
    a = api()
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    o = np.ascontiguousarray(ordering, dtype=np.int32)
    w = np.zeros(n * (n - 1) // 2, dtype=np.float64)
    st = _capi.FnnSwStats()
    rc = a.split_weights_f64(D.ctypes.data_as(C.POINTER(C.c_double)), n, n, o.ctypes.data_as(C.POINTER(C.c_int32)),
                             device, w.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    if rc == -7 and allow_inexact:   # FNN_EINEXACT: the weights are there, their Kuhn-Tucker check is above 1e-9 (stats say by how much)
        rc = 0
    try:
        a.check(rc)
    except _capi.FnnError as e:       # FNN_ECAPACITY leaves the give-up reason, the capacity and the peak of the free set in the stats
        e.stats = {k: getattr(st, k) for k, _ in st._fields_ if not k.startswith(("reserved", "pad_"))}
        raise
    out = {k: getattr(st, k) for k, _ in st._fields_ if not k.startswith(("reserved", "pad_"))}
    out["method"] = ("closed form", "from below", "reference")[st.route]
    out["refactorizations"] = int(st.reserved[1])
    out["solves"] = int(st.reserved[2])
    return w, out


def split_weights_sparse(D: np.ndarray, ordering: np.ndarray, threshold: float = 1e-6, device: int = 0, capacity: int = 0):
    """`fnn_split_weights_sparse_f64`: only the weights above `threshold` (the reference's list, FastNN.java:455-466), as
    (indices[k], weights[k]) in ascending live index.  Returns (indices, weights, stats dict)."""
    from . import api
    a = api()
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    o = np.ascontiguousarray(ordering, dtype=np.int32)
    cap = int(capacity) if capacity > 0 else min(n * (n - 1) // 2, max(64 * n, 4096))
    while True:
        idx = np.zeros(cap, dtype=np.int64)
        w = np.zeros(cap, dtype=np.float64)
        cnt = C.c_int64(0)
        st = _capi.FnnSwStats()
        a.check(a.split_weights_sparse_f64(D.ctypes.data_as(C.POINTER(C.c_double)), n, n, o.ctypes.data_as(C.POINTER(C.c_int32)), device,
                                           float(threshold), idx.ctypes.data_as(C.POINTER(C.c_int64)), w.ctypes.data_as(C.POINTER(C.c_double)),
                                           cap, C.byref(cnt), C.byref(st)))
        if cnt.value <= cap:
            break
        cap = int(cnt.value)   # (more splits than room: once more with exactly enough)
    out = {k: getattr(st, k) for k, _ in st._fields_ if not k.startswith(("reserved", "pad_"))}
    out["method"] = ("closed form", "from below", "reference")[st.route]
    return idx[:cnt.value], w[:cnt.value], out
