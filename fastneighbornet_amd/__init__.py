"""fastneighbornet_amd -- MI355X-native Canonical Neighbor-Net agglomeration engine.

The compute path is `libfastnn_hip.so` (hand-written gfx950 HIP kernels behind the
C ABI of include/fastnn.h).  This package is the thin Python host side used by the
tests and bench: it mirrors the reference's seam for the path,
`new NeighborNetCanonical(d, ntax, threads, pool).runNeighborNet()`
(NeighborNetCanonical.java:34-36, NetMakerOriginal.java:129), over ctypes.
There is no CPU fallback: if the library is missing or no HIP device is usable the
calls raise.
"""
from __future__ import annotations

import ctypes as _C
import os as _os

from . import _capi
from ._capi import FnnError, Handle  # noqa: F401

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.path.join(_HERE, "libfastnn_hip.so")
_api = None
TORCH_LOADED_FIRST = False


def api() -> _capi.Api:
    """Load libfastnn_hip.so (built in-tree by `python -m fastneighbornet_amd.build`)."""
    global _api
    if _api is None:
        if not _os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m fastneighbornet_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        import sys as _sys
        global TORCH_LOADED_FIRST
        # PyTorch bundles its own HIP runtime; whichever of the two is loaded first serves both
        # (same soname).  distributed.rccl_path() picks the librccl that matches.
        TORCH_LOADED_FIRST = "torch" in _sys.modules
        lib = _C.CDLL(LIB_PATH)
        a = _capi.Api(lib, "fnn_")
        a._fn("abi_version", _C.c_int32, [])
        a._fn("device_count", _C.c_int32, [])
        a._fn("set_matrix_device", _C.c_int32, [_C.c_void_p, _C.c_void_p, _C.c_int64])
        a._fn("set_scan_timing", _C.c_int32, [_C.c_void_p, _C.c_int32])
        a._fn("get_kernel_times", _C.c_int32, [_C.c_void_p, _C.POINTER(_C.c_double), _C.POINTER(_C.c_int64)])
        a._fn("get_exchange_times", _C.c_int32, [_C.c_void_p, _C.POINTER(_C.c_double), _C.POINTER(_C.c_int64)])
        a._fn("canonical_order_f64", _C.c_int32,
              [_C.POINTER(_C.c_double), _C.c_int32, _C.c_int64, _C.POINTER(_capi.FnnOpts),
               _C.POINTER(_C.c_int32), _C.POINTER(_capi.FnnStats)])
        a._fn("test_chain_sum", _C.c_int32,
              [_C.c_int32, _C.POINTER(_C.c_double), _C.c_int32, _C.c_int32, _C.c_int32,
               _C.POINTER(_C.c_double), _C.POINTER(_C.c_int32)])
        a._fn("comm_unique_id", _C.c_int32, [_C.POINTER(_C.c_uint8), _C.c_char_p])
        a._fn("comm_probe", _C.c_int32, [_C.c_char_p])
        a._fn("comm_init_rccl", _C.c_int32, [_C.c_void_p, _C.c_int32, _C.c_int32, _C.POINTER(_C.c_uint8), _C.c_char_p])
        a._fn("stream_probe", _C.c_int32, [_C.c_int32, _C.c_int64, _C.c_int32, _C.POINTER(_C.c_double)])
        a._fn("split_weights_f64", _C.c_int32,
              [_C.POINTER(_C.c_double), _C.c_int32, _C.c_int64, _C.POINTER(_C.c_int32), _C.c_int32,
               _C.POINTER(_C.c_double), _C.POINTER(_capi.FnnSwStats)])
        a._fn("split_weights_release_cache", _C.c_int32, [])
        a._fn("split_weights_sparse_f64", _C.c_int32,
              [_C.POINTER(_C.c_double), _C.c_int32, _C.c_int64, _C.POINTER(_C.c_int32), _C.c_int32, _C.c_double,
               _C.POINTER(_C.c_int64), _C.POINTER(_C.c_double), _C.c_int64, _C.POINTER(_C.c_int64), _C.POINTER(_capi.FnnSwStats)])
        _api = a
    return _api


from .canonical import NeighborNetCanonical, NeighborNetLocal, canonical_order, split_weights, split_weights_sparse  # noqa: E402,F401
