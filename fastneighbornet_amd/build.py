"""Builds libfastnn_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libfastnn_hip.so")
SRC = os.path.join(HERE, "csrc", "fnn_hip.hip")
SRC_SPLITS = os.path.join(HERE, "csrc", "fnn_splits.hip")  # circular split weights
DEPS = [SRC, SRC_SPLITS, os.path.join(HERE, "csrc", "fnn_core.h"), os.path.join(HERE, "csrc", "fnn_engine.h"),
        os.path.join(HERE, "csrc", "fnn_chain.h"), os.path.join(ROOT, "include", "fastnn.h")]

# -ffp-contract=off: one rounding per fp64 operation on host and device (parity with Java doubles)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-parameter"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libfastnn_hip.so cannot be built (no CPU fallback exists)")


def check_isa(asm: str, src: str) -> None:
    """hipcc (ROCm 7.2, LLVM 22 git) was seen to lower a uniform `cond ? a : b` whose
    condition is a VALU compare of uniform values held in VGPRs to `v_cmp` + `s_cselect`
    without moving VCC into SCC, so the select read a stale SCC.  Scan the device ISA (kept by
    -save-temps from the one compilation of each source) for an s_cselect / s_cbranch_scc whose SCC
    producer is not a compare and refuse the build."""
    import re
    writers = re.compile(r"^\s*(s_cmp|s_bitcmp|s_cmpk|s_add|s_sub|s_addc|s_subb|s_and|s_or|s_xor|s_not|s_lshl|"
                         r"s_lshr|s_ashr|s_min|s_max|s_abs|s_andn2|s_orn2|s_nand|s_nor|s_xnor|s_bfe|"
                         r"s_absdiff|s_wqm|s_quadmask|s_bcnt|s_ff|s_flbit|s_addk)")
    cmp_like = re.compile(r"^\s*(s_cmp|s_bitcmp|s_cmpk|s_and_b64|s_or_b64|s_andn2_b64|s_and_b32|s_or_b32|"
                          r"s_xor_b64|s_orn2_b64)")
    # signature of the miscompile: a VALU compare (result in VCC) sits between the last SCC
    # writer and an SCC consumer, i.e. the select was meant to test that compare.  (An
    # s_cselect right after s_add_u32 / s_addc_u32 legitimately captures the carry.)
    last, last_i, vcmp_i, kern, bad = None, 0, -1, None, []
    for i, line in enumerate(open(asm), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, last, last_i, vcmp_i = m.group(1), None, 0, -1
        t = line.strip()
        if t.startswith("v_cmp"):
            vcmp_i = i
        if t.startswith("s_cselect") or t.startswith("s_cbranch_scc"):
            if (last is None or not cmp_like.match(last)) and vcmp_i > last_i:
                bad.append(f"{kern}:{i}: {t} (SCC from: {last.strip() if last else None}; v_cmp at line {vcmp_i})")
        if writers.match(line):
            last, last_i = line, i
    if bad:
        raise RuntimeError(f"suspicious SCC use in device ISA of {os.path.basename(src)} (compiler miscompile?):\n" + "\n".join(bad))


def compile_one(src: str, workdir: str) -> str:
    """One compilation per source: object file + (through -save-temps) the device ISA that check_isa reads."""
    obj = os.path.join(workdir, os.path.splitext(os.path.basename(src))[0] + ".o")
    r = subprocess.run([hipcc()] + [f for f in FLAGS if f != "-shared"] + ["-c", "-save-temps", "-o", obj, src], cwd=workdir,
                       stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    stem = os.path.splitext(os.path.basename(src))[0]
    asm = [f for f in os.listdir(workdir) if f.startswith(stem + "-hip-amdgcn") and f.endswith(".s")]
    if not asm:
        raise RuntimeError(f"no device ISA kept for {src}")
    check_isa(os.path.join(workdir, asm[0]), src)
    return obj


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libfastnn_host.so")
CLI = os.path.join(HERE, "bin", "fastnn")


def build_host(force: bool = False) -> None:
    """C++ host side: Phylip reader / formatting library (g++) and the `fastnn` CLI."""
    srcs = [os.path.join(HOST_DIR, f) for f in ("fastnn_host.cpp", "fastnn_host.hpp", "fastnn_main.cpp")]
    newest = max(os.path.getmtime(p) for p in srcs + [os.path.join(ROOT, "include", "fastnn.h")])
    if force or not os.path.exists(HOST_LIB) or os.path.getmtime(HOST_LIB) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-o", HOST_LIB,
                               os.path.join(HOST_DIR, "fastnn_host.cpp")])
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(newest, os.path.getmtime(LIB)):
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        subprocess.check_call([hipcc(), "-O2", "-std=c++17", "-pthread", "-o", CLI, os.path.join(HOST_DIR, "fastnn_main.cpp"),
                               os.path.join(HOST_DIR, "fastnn_host.cpp"), "-L" + HERE, "-lfastnn_hip",
                               "-Wl,-rpath,$ORIGIN/.."])


OBJ_DIR = os.path.join(HERE, "csrc", "build")  # objects are kept (git-ignored): a source is recompiled only when it or a header changed
HEADERS = [os.path.join(HERE, "csrc", "fnn_core.h"), os.path.join(HERE, "csrc", "fnn_engine.h"), os.path.join(HERE, "csrc", "fnn_chain.h"),
           os.path.join(ROOT, "include", "fastnn.h")]


def toolchain_stamp() -> str:
    """What the kept objects were compiled WITH: flags + compiler identity.  A change of either recompiles everything
    (and thereby re-runs check_isa on the new compiler's output); mtimes alone would reuse stale objects."""
    import hashlib
    try:
        ver = subprocess.run([hipcc(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    except OSError as e:  # pragma: no cover
        ver = repr(e)
    return hashlib.sha256((" ".join(FLAGS) + "\n" + ver).encode()).hexdigest()


def build(force: bool = False) -> str:
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp_file, stamp = os.path.join(OBJ_DIR, "toolchain.stamp"), toolchain_stamp()
    if not (os.path.exists(stamp_file) and open(stamp_file).read().strip() == stamp):
        # Objects without a matching stamp: other flags, another hipcc / ROCm, or a tree from before the stamp existed.  On a
        # box that received prebuilt objects together with their stamp (the GPU box: same image) nothing is recompiled.
        force = force or any(os.path.exists(os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(x))[0] + ".o")) for x in (SRC, SRC_SPLITS)) and \
            os.path.exists(stamp_file)
    todo, objs = [], []
    for src in (SRC, SRC_SPLITS):
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(p) for p in [src] + HEADERS)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            todo.append((src, obj))
    if todo:
        # sources compile side by side (fnn_splits.hip's fully unrolled 64 x 64 inverse-Cholesky kernel takes two minutes)
        with tempfile.TemporaryDirectory() as td:
            def one(job):
                src, obj = job
                d = os.path.join(td, os.path.splitext(os.path.basename(src))[0])
                os.makedirs(d)
                shutil.copyfile(compile_one(src, d), obj)
            with ThreadPoolExecutor(2) as ex:
                list(ex.map(one, todo))
    if todo or not os.path.exists(stamp_file):
        open(stamp_file, "w").write(stamp + "\n")
    if todo or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        # (rocBLAS: the plain fp64 GEMM / GEMV / SYRK calls of the split-weight solver)
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lrocblas"])
    build_host(force)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
