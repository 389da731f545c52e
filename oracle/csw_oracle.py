"""ctypes binding of the test-only split-weight oracle (oracle/csw_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see csw_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libcsw_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "csw_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-Wall", "-Wextra", "-ffp-contract=off",
                               "-fno-fast-math", "-shared", "-o", _LIB_PATH, src, "-lm"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_long)
        L.csw_calculate_atx.argtypes = [C.c_int, dp, dp]
        L.csw_calculate_ab.argtypes = [C.c_int, dp, dp]
        L.csw_unconstrained_ls.argtypes = [C.c_int, dp, dp]
        L.csw_active_conjugate.argtypes = [C.c_int, dp, dp, lp]
        L.csw_setup_d.argtypes = [C.c_int, dp, ip, dp]
        L.csw_to_live.argtypes = [C.c_int, dp, dp]
        L.csw_split_weights.argtypes = [C.c_int, dp, ip, dp, lp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def npairs(n):
    return n * (n - 1) // 2


def calculate_atx(n, y):
    y = np.ascontiguousarray(y, np.float64)
    p = np.zeros(npairs(n))
    lib().csw_calculate_atx(n, _dp(y), _dp(p))
    return p


def calculate_ab(n, b):
    b = np.ascontiguousarray(b, np.float64)
    d = np.zeros(npairs(n))
    lib().csw_calculate_ab(n, _dp(b), _dp(d))
    return d


def unconstrained_ls(n, d):
    d = np.ascontiguousarray(d, np.float64)
    x = np.zeros(npairs(n))
    lib().csw_unconstrained_ls(n, _dp(d), _dp(x))
    return x


def setup_d(D, ordering):
    D = np.ascontiguousarray(D, np.float64)
    n = D.shape[0]
    o = np.ascontiguousarray(ordering, np.int32)
    d = np.zeros(npairs(n))
    lib().csw_setup_d(n, _dp(D), o.ctypes.data_as(C.POINTER(C.c_int32)), _dp(d))
    return d


def split_weights(D, ordering):
    """Non-negative least-squares weights of the circular splits, live path index order
    (FastNN.java:405-419): k runs over (i, j), 0 <= i < j <= n-1, split = taxa ordering[i+1 .. j]."""
    D = np.ascontiguousarray(D, np.float64)
    n = D.shape[0]
    o = np.ascontiguousarray(ordering, np.int32)
    live = np.zeros(npairs(n))
    st = (C.c_long * 3)()
    lib().csw_split_weights(n, _dp(D), o.ctypes.data_as(C.POINTER(C.c_int32)), _dp(live), st)
    return live, tuple(st)


def live_design_matrix(n, ordering):
    """The dense design matrix of the reference's live path (FastNN.java:405-437): rows = taxon pairs
    (1,2),(1,3),... in the original labelling, columns = splits (i, j) in the live order."""
    pairs = [(a, b) for a in range(1, n + 1) for b in range(a + 1, n + 1)]
    cols = []
    for i in range(n):
        for j in range(i + 1, n):
            cols.append(set(int(t) for t in ordering[i + 1: j + 1]))
    A = np.zeros((len(pairs), len(cols)))
    for k, S in enumerate(cols):
        for r, (a, b) in enumerate(pairs):
            A[r, k] = 1.0 if ((a in S) != (b in S)) else 0.0
    return A


def packed_distances(D):
    n = D.shape[0]
    return np.array([D[a, b] for a in range(n) for b in range(a + 1, n)])
