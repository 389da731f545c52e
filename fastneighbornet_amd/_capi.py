"""ctypes view of the C ABI declared in include/fastnn.h.

`Api(lib, prefix)` binds one shared library.  The product uses
`Api(libfastnn_hip.so, "fnn_")`; the CPU tests reuse the same class for the
emulation driver (`tests/emu`, prefix "emu_"), which exports the same entry
points over the same host logic.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

FNN_OK = 0
STATUS_NAMES = {0: "FNN_OK", -1: "FNN_EINVAL", -2: "FNN_ENOMEM", -3: "FNN_EHIP", -4: "FNN_ERCCL",
                -5: "FNN_ESTATE", -6: "FNN_ECAPACITY", -7: "FNN_EINEXACT"}

KIND_2WAY, KIND_3WAY, KIND_4WAY, KIND_FINISH = 2, 3, 4, 5


class FnnOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("validate", C.c_int32), ("record_events", C.c_int32),
                ("force_exact_rx", C.c_int32), ("disable_screen", C.c_int32), ("lookahead", C.c_int32),
                ("lookahead_pairs", C.c_int32), ("mode", C.c_int32), ("relaxed_seed_lo", C.c_uint32),
                ("relaxed_seed_hi", C.c_uint32), ("relaxed_min_active", C.c_int32), ("reserved", C.c_int32 * 5)]


class FnnEvent(C.Structure):
    _fields_ = [("m_before", C.c_int32), ("c_before", C.c_int32), ("cx_id", C.c_int32),
                ("cy_id", C.c_int32), ("x_id", C.c_int32), ("y_id", C.c_int32), ("kind", C.c_int32),
                ("u_id", C.c_int32), ("best", C.c_double), ("entries", C.c_int64)]

    def key(self):
        return (self.m_before, self.c_before, self.cx_id, self.cy_id, self.x_id, self.y_id,
                self.kind, self.u_id)


class FnnStats(C.Structure):
    _fields_ = [("n_events", C.c_int64), ("sum_entries", C.c_int64), ("t_init_s", C.c_double),
                ("t_agglom_s", C.c_double), ("t_expand_s", C.c_double), ("t_total_s", C.c_double),
                ("t_scan_s", C.c_double), ("scan_launches", C.c_int64), ("scan_bytes", C.c_int64),
                ("n_rx_certified", C.c_int64), ("n_rx_exact", C.c_int64), ("n_screen_events", C.c_int64),
                ("n_rescan_units", C.c_int64), ("n_base_scans", C.c_int64), ("n_window_hits", C.c_int64),
                ("n_window_fails", C.c_int64), ("window_pairs", C.c_int64), ("bytes_total", C.c_int64),
                ("n_handover_retries", C.c_int64), ("n_sweeps_exact", C.c_int64), ("t_plain_s", C.c_double),
                ("plain_launches", C.c_int64), ("plain_bytes", C.c_int64), ("n_stalled_events", C.c_int64),
                ("n_relaxed_events", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class FnnSwStats(C.Structure):
    _fields_ = [("outer_iterations", C.c_int64), ("cg_calls", C.c_int64), ("cg_iterations", C.c_int64),
                ("nsplits", C.c_int64), ("t_solve_s", C.c_double), ("reserved", C.c_int64 * 3),
                # ABI version 2
                ("route", C.c_int32), ("certified", C.c_int32), ("kkt_violation", C.c_double), ("final_threshold_rel", C.c_double),
                ("n_set_aside", C.c_int64), ("capacity", C.c_int64), ("free_set_peak", C.c_int64), ("giveup_reason", C.c_int32),
                ("pad_", C.c_int32), ("entered", C.c_int64), ("screened_out", C.c_int64), ("departed", C.c_int64),
                ("t_alloc_s", C.c_double), ("reserved2", C.c_int64 * 4)]


EVENT_DTYPE = np.dtype(
    [("m_before", "<i4"), ("c_before", "<i4"), ("cx_id", "<i4"), ("cy_id", "<i4"),
     ("x_id", "<i4"), ("y_id", "<i4"), ("kind", "<i4"), ("u_id", "<i4"),
     ("best", "<f8"), ("entries", "<i8")], align=True)


# int32_t (*fnn_allgather_fn)(void* ctx, const void* send, void* recv, int32_t bytes_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)


class FnnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Api:
    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib = lib
        self.prefix = prefix
        f = self._fn
        f("last_error", C.c_char_p, [])
        f("create", C.c_int32, [C.c_int32, C.POINTER(FnnOpts), C.POINTER(C.c_void_p)])
        f("destroy", C.c_int32, [C.c_void_p])
        f("set_rows", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _dp, C.c_int64])
        f("set_packed_upper", C.c_int32, [C.c_void_p, _dp])
        f("synth", C.c_int32, [C.c_void_p, C.c_uint64, C.c_int32])
        f("run", C.c_int32, [C.c_void_p, _ip, C.POINTER(FnnStats)])
        f("begin", C.c_int32, [C.c_void_p])
        f("step", C.c_int32, [C.c_void_p, C.POINTER(FnnEvent)])
        f("finish", C.c_int32, [C.c_void_p, _ip])
        f("get_events", C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64])
        f("get_counts", C.c_int32, [C.c_void_p, _ip, _ip, _ip])
        f("get_nodes", C.c_int32, [C.c_void_p, _ip, _ip, _dp])
        f("get_live_matrix", C.c_int32, [C.c_void_p, _dp])
        f("get_matrix", C.c_int32, [C.c_void_p, _dp, C.c_int64])
        f("comm_init_host", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, ALLGATHER_FN, C.c_void_p])

    def _fn(self, name, restype, argtypes, optional=False):
        try:
            fn = getattr(self.lib, self.prefix + name)
        except AttributeError:
            if optional:
                return None
            raise
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, name, fn)
        return fn

    def check(self, rc):
        if rc < 0:
            raise FnnError(rc, (self.last_error() or b"").decode("utf-8", "replace"))
        return rc


class Handle:
    """Thin RAII wrapper over an engine handle of either library."""

    def __init__(self, api: Api, n: int, device: int = 0, validate: bool = False,
                 record_events: bool = False, force_exact_rx: bool = False, disable_screen: bool = False,
                 lookahead: int = 0, lookahead_pairs: int = 0, relaxed_seed=None, relaxed_min_active: int = 0):
        self.api = api
        self.n = int(n)
        opts = FnnOpts()
        opts.device = device
        opts.validate = 1 if validate else 0
        opts.record_events = 1 if record_events else 0
        opts.force_exact_rx = 1 if force_exact_rx else 0
        opts.disable_screen = 1 if disable_screen else 0
        opts.lookahead = lookahead
        opts.lookahead_pairs = lookahead_pairs
        if relaxed_seed is not None:  # -mode Relaxed (NeighborNetLocal) with java.util.Random(relaxed_seed)
            opts.mode = 1
            opts.relaxed_seed_lo = int(relaxed_seed) & 0xFFFFFFFF
            opts.relaxed_seed_hi = (int(relaxed_seed) >> 32) & 0xFFFFFFFF
            opts.relaxed_min_active = relaxed_min_active
        h = C.c_void_p()
        api.check(api.create(self.n, C.byref(opts), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.api.destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- several ranks --
    def comm_init_host(self, world: int, rank: int, allgather):
        """Test transport: `allgather(send: bytes) -> list[bytes]` (one entry per rank)."""
        def _cb(ctx, send, recv, nbytes):
            try:
                parts = allgather(C.string_at(send, nbytes))
                C.memmove(recv, b"".join(parts), nbytes * world)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLGATHER_FN(_cb)  # keep alive
        self.api.check(self.api.comm_init_host(self._h, world, rank, self._cb, None))

    def comm_init_rccl(self, world: int, rank: int, uid: bytes, rccl_path: str | None = None):
        p = rccl_path.encode() if rccl_path else None
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self.api.check(self.api.comm_init_rccl(self._h, world, rank, buf, p))

    # -- matrix --
    def set_matrix(self, D: np.ndarray, chunk_rows: int = 0):
        D = np.ascontiguousarray(D, dtype=np.float64)
        if D.shape != (self.n, self.n):
            raise ValueError(f"matrix must be {self.n} x {self.n}")
        step = chunk_rows if chunk_rows > 0 else max(self.n, 1)
        for r0 in range(0, self.n, step):
            cnt = min(step, self.n - r0)
            self.api.check(self.api.set_rows(self._h, r0, cnt, D[r0:].ctypes.data_as(_dp), self.n))

    def set_packed_upper(self, packed: np.ndarray):
        """DistancesAndNames.distances: the strict upper triangle, row-major (DistancesAndNames.java:24-38)."""
        packed = np.ascontiguousarray(packed, dtype=np.float64)
        if packed.shape != (self.n * (self.n - 1) // 2,):
            raise ValueError(f"packed triangle must have {self.n * (self.n - 1) // 2} entries")
        self.api.check(self.api.set_packed_upper(self._h, packed.ctypes.data_as(_dp)))

    def synth(self, seed: int, dist: str = "uniform53"):
        self.api.check(self.api.synth(self._h, seed, {"uniform53": 0, "dec4": 1}[dist]))

    # -- run --
    def run(self):
        order = np.zeros(self.n + 1, dtype=np.int32)
        st = FnnStats()
        self.api.check(self.api.run(self._h, order.ctypes.data_as(_ip), C.byref(st)))
        return order, st

    def begin(self):
        self.api.check(self.api.begin(self._h))

    def step(self):
        ev = FnnEvent()
        r = self.api.check(self.api.step(self._h, C.byref(ev)))
        return ev if r == 1 else None

    def finish(self):
        order = np.zeros(self.n + 1, dtype=np.int32)
        self.api.check(self.api.finish(self._h, order.ctypes.data_as(_ip)))
        return order

    def events(self) -> np.ndarray:
        k = self.api.get_events(self._h, None, 0)
        out = np.zeros(max(k, 1), dtype=EVENT_DTYPE)
        self.api.get_events(self._h, out.ctypes.data, k)
        return out[:k]

    def counts(self):
        m, c, nn = C.c_int32(), C.c_int32(), C.c_int32()
        self.api.check(self.api.get_counts(self._h, C.byref(m), C.byref(c), C.byref(nn)))
        return m.value, c.value, nn.value

    def nodes(self):
        m, _, _ = self.counts()
        ids = np.zeros(max(self.n, 1), np.int32)
        nbr = np.zeros(max(self.n, 1), np.int32)
        sx = np.zeros(max(self.n, 1), np.float64)
        self.api.check(self.api.get_nodes(self._h, ids.ctypes.data_as(_ip), nbr.ctypes.data_as(_ip),
                                          sx.ctypes.data_as(_dp)))
        return ids[:m], nbr[:m], sx[:m]

    def matrix(self) -> np.ndarray:
        """The resident n x n matrix (after set_matrix / synth, before run consumes it) - fnn_get_matrix."""
        out = np.empty((self.n, self.n), dtype=np.float64)
        self.api.check(self.api.get_matrix(self._h, out.ctypes.data_as(_dp), self.n))
        return out

    def live_matrix(self) -> np.ndarray:
        m, _, _ = self.counts()
        out = np.zeros((m, m), dtype=np.float64)
        self.api.check(self.api.get_live_matrix(self._h, out.ctypes.data_as(_dp)))
        return out
