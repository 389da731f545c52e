"""CPU checks of the drop-in boundary: the shared library loads without a GPU and
exports every symbol include/fastnn.h declares; no compute is attempted."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "fastnn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fnn_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("fnn_create", "fnn_set_rows", "fnn_run", "fnn_step", "fnn_destroy",
                 "fnn_canonical_order_f64", "fnn_last_error"):
        assert must in names


def test_library_builds_loads_and_exports_all_symbols():
    from fastneighbornet_amd import build
    lib = C.CDLL(build.build())
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in fastnn.h but not exported"
    lib.fnn_abi_version.restype = C.c_int32
    assert lib.fnn_abi_version() == 2


def test_fails_loudly_without_device():
    """No CPU fallback: on a box without a HIP device create() must fail with FNN_EHIP."""
    import fastneighbornet_amd as fa
    from fastneighbornet_amd._capi import FnnError, Handle
    a = fa.api()
    if a.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(FnnError) as ei:
        Handle(a, 16)
    assert ei.value.code == -3
    import numpy as np
    with pytest.raises(FnnError):
        fa.canonical_order(np.zeros((8, 8)))


def test_struct_layouts_match_header():
    from fastneighbornet_amd import _capi
    assert C.sizeof(_capi.FnnEvent) == 48
    assert C.sizeof(_capi.FnnOpts) == 64
    assert C.sizeof(_capi.FnnStats) == 25 * 8  # 18 named 8-byte fields + reserved[7]
    assert _capi.EVENT_DTYPE.itemsize == 48
    assert C.sizeof(_capi.FnnSwStats) == 8 * 8 + 2 * 4 + 2 * 8 + 3 * 8 + 2 * 4 + 3 * 8 + 8 + 4 * 8  # fnn_sw_stats, ABI version 2


def test_oracle_is_test_infrastructure_only():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/: no file of
    the product (the package, its C/C++/HIP sources, the host mirror, the tools) mentions it in code."""
    offenders = []
    for base in ("fastneighbornet_amd", "include", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c", ".sh")):
                    continue
                txt = open(os.path.join(dp, f), errors="replace").read()
                for line in txt.split("\n"):
                    code = line.split("#")[0] if f.endswith((".py", ".sh")) else line.split("//")[0]
                    if re.search(r"(from|import)\s+oracle\b|oracle/|nnet_oracle|csw_oracle", code):
                        offenders.append(f"{os.path.relpath(os.path.join(dp, f), ROOT)}: {line.strip()[:80]}")
    assert not offenders, offenders
    # bench.py: the oracle only inside the cpu_baseline function; __graft_entry__: only build() (compiling
    # the checker) and smoke()
    b = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"nnet_oracle|csw_oracle|from oracle|import oracle", b)]
    assert uses, "bench.py lost its cpu_baseline leg"
    spans = []
    for fn in ("def cpu_baseline", "def cpu_measured_small"):  # the two functions of the cpu_baseline leg
        start = b.index(fn)
        spans.append((start, b.index("\ndef ", start + 1)))
    assert all(any(s0 < u < s1 for s0, s1 in spans) for u in uses), "bench.py uses the oracle outside its cpu_baseline leg"
