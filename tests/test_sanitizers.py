"""CPU sanitizer leg ("race detection / sanitizers"): the emulation of the engine's per-thread
bodies + host logic (tests/emu/fnn_emu.cpp over fnn_core.h / fnn_engine.h / fnn_chain.h) and the C++ host side
(Phylip reader, Java number formatting, Nexus writer) are built with -fsanitize=address,undefined and a part of the
CPU suite runs again on those builds in a child process (libasan has to be preloaded into the interpreter).
GPU AddressSanitizer is not available on the pool; this covers everything that compiles for the host."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "build")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _newer(out, srcs):
    return not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in srcs)


@pytest.fixture(scope="module")
def san_libs():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("gcc has no libasan here")
    os.makedirs(BUILD, exist_ok=True)
    csrc = os.path.join(ROOT, "fastneighbornet_amd", "csrc")
    emu = os.path.join(BUILD, "libfnn_emu_asan.so")
    esrc = [os.path.join(ROOT, "tests", "emu", "fnn_emu.cpp")] + [os.path.join(csrc, f) for f in ("fnn_core.h", "fnn_engine.h", "fnn_chain.h")]
    if _newer(emu, esrc):
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas"] + SAN + ["-o", emu, esrc[0]])
    host = os.path.join(BUILD, "libfastnn_host_asan.so")
    hdir = os.path.join(ROOT, "fastneighbornet_amd", "host")
    hsrc = [os.path.join(hdir, "fastnn_host.cpp"), os.path.join(hdir, "fastnn_host.hpp")]
    if _newer(host, hsrc):
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared"] + SAN + ["-o", host, hsrc[0]])
    return rt, emu, host


def _run(rt, env_extra, args):
    # (libstdc++ is preloaded too: the ASan runtime resolves __cxa_throw when it starts, and the interpreter itself
    #  does not link the C++ runtime - without this the first C++ exception inside an instrumented library aborts)
    cxx = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=rt + (" " + cxx if os.path.isabs(cxx) else ""), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "passed" in r.stdout


def test_emulation_under_asan_ubsan(san_libs):
    rt, emu, _ = san_libs
    # whole trajectories against the oracle in all thread orders (slot bookkeeping, windows, the block form of the
    # involved slots, the cached slot tables, T), the chain-sum model, two ranks' worth of exchange logic in-process
    _run(rt, {"FNN_EMU_LIB": emu}, ["tests/test_emu_parity.py", "tests/test_chain_sum.py", "-k", "not big"])
    # the Relaxed mode's search (permutation, per-slot row-minimum cache, tie lists) against the oracle
    _run(rt, {"FNN_EMU_LIB": emu}, ["tests/test_relaxed.py", "-m", "not gpu", "-k", "emulation or rejects"])


def test_host_side_under_asan_ubsan(san_libs):
    rt, _, host = san_libs
    _run(rt, {"FNN_HOST_LIB": host}, ["tests/test_host_cli.py", "-m", "not gpu", "-k", "reader or formatting or parallel_writer"])
