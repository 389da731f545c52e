import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The test-only CPU oracle (oracle/nnet_oracle.c), built on demand with gcc."""
    from oracle import nnet_oracle
    nnet_oracle.lib()
    return nnet_oracle


@pytest.fixture(scope="session")
def emu_api():
    """CPU emulation of the engine's per-thread bodies (tests/emu) -- test infrastructure."""
    import ctypes as C
    from fastneighbornet_amd._capi import Api
    here = os.path.join(ROOT, "tests", "emu")
    out = os.path.join(here, "build", "libfnn_emu.so")
    if os.environ.get("FNN_EMU_LIB"):  # (the sanitizer leg points at its instrumented build)
        lib = C.CDLL(os.environ["FNN_EMU_LIB"])
        lib.emu_set_order_mode.argtypes = [C.c_int32]
        api = Api(lib, "emu_")
        api.set_order_mode = lib.emu_set_order_mode
        return api
    srcs = [os.path.join(here, "fnn_emu.cpp"),
            os.path.join(ROOT, "fastneighbornet_amd", "csrc", "fnn_core.h"),
            os.path.join(ROOT, "fastneighbornet_amd", "csrc", "fnn_engine.h"),
            os.path.join(ROOT, "include", "fastnn.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                               "-Wno-unknown-pragmas", "-o", out, srcs[0]])
    lib = C.CDLL(out)
    lib.emu_set_order_mode.argtypes = [C.c_int32]
    api = Api(lib, "emu_")
    api.set_order_mode = lib.emu_set_order_mode
    return api


@pytest.fixture(scope="session")
def hip_api():
    """The product library through its C ABI (GPU tests only)."""
    import fastneighbornet_amd as fa
    return fa.api()
