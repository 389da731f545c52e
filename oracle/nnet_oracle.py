"""ctypes binding of the test-only CPU oracle (oracle/nnet_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see nnet_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libnnet_oracle.so")

KIND_2WAY, KIND_3WAY, KIND_4WAY, KIND_FINISH = 2, 3, 4, 5


class Event(C.Structure):
    _fields_ = [
        ("m_before", C.c_int32),
        ("c_before", C.c_int32),
        ("cx_id", C.c_int32),
        ("cy_id", C.c_int32),
        ("x_id", C.c_int32),
        ("y_id", C.c_int32),
        ("kind", C.c_int32),
        ("u_id", C.c_int32),
        ("best", C.c_double),
        ("entries", C.c_int64),
    ]

    def key(self):
        return (self.m_before, self.c_before, self.cx_id, self.cy_id, self.x_id, self.y_id,
                self.kind, self.u_id)


EVENT_DTYPE = np.dtype(
    [("m_before", "<i4"), ("c_before", "<i4"), ("cx_id", "<i4"), ("cy_id", "<i4"),
     ("x_id", "<i4"), ("y_id", "<i4"), ("kind", "<i4"), ("u_id", "<i4"),
     ("best", "<f8"), ("entries", "<i8")], align=True)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "nnet_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "nnet_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.nno_create.restype = C.c_void_p
        L.nno_create.argtypes = [dp, C.c_int32, C.c_int32]
        L.nno_destroy.argtypes = [C.c_void_p]
        L.nno_step.restype = C.c_int32
        L.nno_step.argtypes = [C.c_void_p, C.POINTER(Event)]
        for f in ("nno_num_active", "nno_num_clusters", "nno_num_nodes"):
            getattr(L, f).restype = C.c_int32
            getattr(L, f).argtypes = [C.c_void_p]
        L.nno_get_nodes.argtypes = [C.c_void_p, ip, ip, ip, dp]
        L.nno_matrix.restype = dp
        L.nno_matrix.argtypes = [C.c_void_p]
        L.nno_expand.restype = C.c_int32
        L.nno_expand.argtypes = [C.c_void_p, ip]
        L.nno_run.restype = C.c_int32
        L.nno_run.argtypes = [dp, C.c_int32, C.c_int32, ip, C.c_void_p, C.c_int64,
                              C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.nno_synth.argtypes = [dp, C.c_int32, C.c_uint64, C.c_int32]
        L.nno_set_relaxed.argtypes = [C.c_void_p, C.c_uint64, C.c_int32]
        L.nno_run_relaxed.restype = C.c_int32
        L.nno_run_relaxed.argtypes = [dp, C.c_int32, C.c_uint64, C.c_int32, ip, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.nno_java_random_next_int.restype = C.c_int32
        L.nno_java_random_next_int.argtypes = [C.POINTER(C.c_uint64), C.c_int32, C.c_int32]
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def synth(n: int, seed: int, dist: str = "uniform53") -> np.ndarray:
    """Symmetric synthetic distance matrix."""
    D = np.empty((n, n), dtype=np.float64)
    lib().nno_synth(_dptr(D), n, seed, {"uniform53": 0, "dec4": 1}[dist])
    return D


def run(D: np.ndarray, threads: int = 1, want_events: bool = True):
    """Whole Canonical run. Returns (order[n+1], events structured array, sum_entries)."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    order = np.zeros(n + 1, dtype=np.int32)
    cap = max(n, 1)
    ev = np.zeros(cap if want_events else 1, dtype=EVENT_DTYPE)
    assert EVENT_DTYPE.itemsize == C.sizeof(Event)
    nev = C.c_int64(0)
    se = C.c_int64(0)
    rc = lib().nno_run(_dptr(D), n, threads, _iptr(order), ev.ctypes.data if want_events else None,
                       cap, C.byref(nev), C.byref(se))
    if rc != 0:
        raise RuntimeError("oracle failed")
    return order, ev[: nev.value] if want_events else None, se.value


def run_relaxed(D: np.ndarray, seed: int, min_active: int = 0):
    """Whole Relaxed run (NeighborNetLocal, additive off) with java.util.Random(seed). Returns (order, events)."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    order = np.zeros(n + 1, dtype=np.int32)
    cap = max(n, 1)
    ev = np.zeros(cap, dtype=EVENT_DTYPE)
    nev = C.c_int64(0)
    rc = lib().nno_run_relaxed(_dptr(D), n, seed, min_active, _iptr(order), ev.ctypes.data, cap, C.byref(nev))
    if rc != 0:
        raise RuntimeError(f"relaxed oracle failed ({rc})")
    return order, ev[: nev.value]


def java_random_ints(seed: int, bounds):
    """java.util.Random(seed).nextInt(b) for b in bounds."""
    st = C.c_uint64(seed)
    out = []
    for i, b in enumerate(bounds):
        out.append(lib().nno_java_random_next_int(C.byref(st), b, 1 if i == 0 else 0))
    return out


class Stepper:
    """Event-by-event access to the oracle state (for trajectory parity tests)."""

    def __init__(self, D: np.ndarray, threads: int = 1, relaxed_seed=None, relaxed_min_active: int = 0):
        D = np.ascontiguousarray(D, dtype=np.float64)
        self.n = D.shape[0]
        self._h = lib().nno_create(_dptr(D), self.n, threads)
        if not self._h:
            raise MemoryError("nno_create failed")
        if relaxed_seed is not None:
            lib().nno_set_relaxed(self._h, relaxed_seed, relaxed_min_active)

    def close(self):
        if self._h:
            lib().nno_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def step(self):
        ev = Event()
        r = lib().nno_step(self._h, C.byref(ev))
        if r < 0:
            raise RuntimeError("oracle step failed")
        return ev if r == 1 else None

    @property
    def num_active(self):
        return lib().nno_num_active(self._h)

    @property
    def num_clusters(self):
        return lib().nno_num_clusters(self._h)

    @property
    def num_nodes(self):
        return lib().nno_num_nodes(self._h)

    def nodes(self):
        """(id, distID, nbr_id, Sx) arrays over positions [0, num_active)."""
        m = self.num_active
        ids = np.zeros(self.n, np.int32)
        dist = np.zeros(self.n, np.int32)
        nbr = np.zeros(self.n, np.int32)
        sx = np.zeros(self.n, np.float64)
        lib().nno_get_nodes(self._h, _iptr(ids), _iptr(dist), _iptr(nbr), _dptr(sx))
        return ids[:m], dist[:m], nbr[:m], sx[:m]

    def matrix(self) -> np.ndarray:
        p = lib().nno_matrix(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n, self.n))

    def expand(self) -> np.ndarray:
        order = np.zeros(self.n + 1, dtype=np.int32)
        if lib().nno_expand(self._h, _iptr(order)) != 0:
            raise RuntimeError("oracle expand failed")
        return order
