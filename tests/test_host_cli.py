"""Host side above the C ABI: the Phylip reader (DistancesAndNames.java:43-132 with its
quirks), the header parse (FastNN.java:269-274) and the CLI surface (FastNN.java:136-268,
:394-397).  The GPU test runs BASELINE.json configs[0] (64-taxa Phylip, -mode Canonical)
end to end through the `fastnn` binary."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fastneighbornet_amd")


def write_phylip(path, D, names=None, style="lower", sep=" "):
    n = D.shape[0]
    with open(path, "w") as f:
        f.write(f"{n}\n")
        for i in range(n):
            name = names[i] if names else f"t{i + 1}"
            cols = range(i) if style == "lower" else range(n)
            vals = [repr(float(D[i, j])) for j in cols]
            # the reference splits on single spaces first, then on tabs inside a token
            f.write(name + (" " + sep.join(vals) if vals else "") + "\n")


@pytest.fixture(scope="module")
def hostlib():
    if os.environ.get("FNN_HOST_LIB"):  # (the sanitizer leg points at its instrumented build)
        lib = C.CDLL(os.environ["FNN_HOST_LIB"])
    else:
        from fastneighbornet_amd import build
        build.build()
        lib = C.CDLL(os.path.join(PKG, "libfastnn_host.so"))
    lib.fnnh_read_taxa_count.argtypes = [C.c_char_p]
    lib.fnnh_read_phylip.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p]
    return lib


def read(hostlib, path, n):
    out = np.zeros((n, n))
    names = C.create_string_buffer(256 * n)
    rc = hostlib.fnnh_read_phylip(path.encode(), n, out.ctypes.data_as(C.POINTER(C.c_double)), names)
    return rc, out, [names.raw[i * 256:(i + 1) * 256].split(b"\0")[0].decode() for i in range(n)]


def test_reader_lower_and_square_and_tabs(hostlib, oracle, tmp_path):
    D = oracle.synth(9, 3)
    for style, sep in (("lower", " "), ("square", " "), ("lower", "\t"), ("square", "  ")):
        p = str(tmp_path / f"{style}{len(sep)}.phy")
        write_phylip(p, D, style=style, sep=sep)
        assert hostlib.fnnh_read_taxa_count(p.encode()) == 9
        rc, out, names = read(hostlib, p, 9)
        assert rc == 0 and (out == D).all() and names[0] == "t1" and names[8] == "t9"


def test_reader_quirks(hostlib, tmp_path):
    # header with embedded whitespace; only the first `row` values of a line are used; the token
    # buffer is not cleared between lines (a short line re-uses the previous line's values);
    # rows beyond numTaxa are ignored
    p = str(tmp_path / "q.phy")
    open(p, "w").write(" 4 \t\nA\nB 1.5 99 98\nC 2.5 3.5 77\nD 4.5\nE 9 9 9 9\n")
    assert hostlib.fnnh_read_taxa_count(p.encode()) == 4
    rc, out, names = read(hostlib, p, 4)
    assert rc == 0 and names == ["A", "B", "C", "D"]
    exp = np.zeros((4, 4))
    exp[1, 0] = 1.5
    exp[2, 0], exp[2, 1] = 2.5, 3.5
    exp[3, 0], exp[3, 1], exp[3, 2] = 4.5, 3.5, 77.0  # stale copy[1], copy[2] from line "C"
    exp = exp + exp.T
    assert (out == exp).all()
    # a number the Java parser rejects
    open(p, "w").write("2\nA\nB x1\n")
    assert read(hostlib, p, 2)[0] == -1


def test_cli_surface_without_gpu(tmp_path):
    exe = os.path.join(PKG, "bin", "fastnn")
    r = subprocess.run([exe, "-help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage: FastNN" in r.stdout and "-distFile <file_location>" in r.stdout
    assert r.stderr.startswith("FastNN Version: 0.3.5\n")
    r = subprocess.run([exe, "-order"], capture_output=True, text=True)
    assert "The program needs a distance file!!" in r.stderr and "usage: FastNN" in r.stdout
    r = subprocess.run([exe, "-distFile", str(tmp_path / "missing.phy"), "-order"], capture_output=True, text=True)
    assert r.stderr.count("FileNotFound") == 1 and r.stdout.startswith("usage: FastNN")
    p = tmp_path / "a.phy"
    p.write_text("3\nA\nB 1\nC 2 3\n")
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "No enum constant nnet.NetMakerOriginal.NMMode.BOGUS" in r.stderr
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "random_n", "-order"], capture_output=True, text=True)
    assert r.returncode == 2 and r.stdout == ""
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "relaxed", "-additive", "-order"], capture_output=True, text=True)
    assert r.returncode == 2 and r.stdout == "" and "-additive" in r.stderr
    # -mode Relaxed (FastNN.java:329-338); ntax <= 3 needs no device
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "relaxed", "-seed", "5", "-order"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == "[0, 1, 2, 3]\n"
    assert "Using the relaxed version without additivity checking.\n" in r.stderr
    # ntax <= 3: identity order without touching a device (NetMakerOriginal.java:133-140)
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "Canonical", "-order", "-time"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == "[0, 1, 2, 3]\n"
    assert "Calculating a tree for 3 taxa using 1 thread(s).\n" in r.stderr
    assert "Using the canonical implementation.\n" in r.stderr and "Got the order in (s): " in r.stderr


@pytest.mark.gpu
def test_cli_config0_64_taxa_phylip(oracle, tmp_path):
    D = oracle.synth(64, 1)
    o_ref, _, _ = oracle.run(D)
    p = str(tmp_path / "c0.phy")
    write_phylip(p, D)
    exe = os.path.join(PKG, "bin", "fastnn")
    r = subprocess.run([exe, "-distFile", p, "-mode", "Canonical", "-threads", "1", "-order", "-time"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "[" + ", ".join(str(int(v)) for v in o_ref) + "]\n"
    assert "Got the order in (s): " in r.stderr
    # without -order: split weights on the GPU + the Nexus document (FastNN.java:398-491)
    # the weights: the optimum of the live path's dense non-negative least-squares problem (FastNN.java:401-454)
    from oracle import csw_oracle as W
    import scipy.optimize as so
    w_ref, _ = so.nnls(W.live_design_matrix(64, o_ref), W.packed_distances(D), maxiter=10 ** 7)
    r = subprocess.run([exe, "-distFile", p, "-mode", "Canonical", "-time"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Got the splits and weights in (s): " in r.stderr and "Wrote the output in (s): " in r.stderr
    txt = r.stdout.split("\n")
    assert txt[0] == "#nexus" and "BEGIN Splits;" in txt
    i = txt.index("BEGIN Splits;")
    ns = int(txt[i + 1].split("nsplits=")[1].rstrip(";"))
    assert txt[i + 4] == "CYCLE " + " ".join(str(int(t)) for t in o_ref[1:]) + ";"
    # every split line against the dense optimum, split by split: the line's taxa identify the live-order index (FastNN.java:405-419)
    n = 64
    live_index = {}
    k = 0
    for a in range(n):
        for b in range(a + 1, n):
            live_index[tuple(sorted(int(t) for t in o_ref[a + 1: b + 1]))] = k
            k += 1
    seen = {}
    for line in txt[i + 6: i + 6 + ns]:
        head, wtxt, taxa = line.split(" \t ")
        seen[live_index[tuple(sorted(int(t) for t in taxa.rstrip(",").split()))]] = float(wtxt)
    assert len(seen) == ns
    near = np.abs(w_ref - 1e-6) < 1e-9          # a weight ON the reference's threshold may fall either side
    for k in range(len(w_ref)):
        if near[k]:
            continue
        assert (k in seen) == (w_ref[k] > 1e-6), (k, w_ref[k], seen.get(k))
    assert not near.any() or abs(ns - int((w_ref > 1e-6).sum())) <= int(near.sum())
    for k, wv in seen.items():
        assert abs(wv - w_ref[k]) <= 1e-6 * max(1.0, w_ref.max()), (k, wv, w_ref[k])


def test_java_double_formatting_and_nexus_document(hostlib, oracle, tmp_path):
    """N2 (OutputPrinter.java:8-96): number formatting follows Double.toString, the document follows
    the reference's block order; splits come from the live-order weights with the 1e-6 threshold."""
    hostlib.fnnh_java_double.argtypes = [C.c_double, C.c_char_p]
    def jd(x):
        b = C.create_string_buffer(32)
        hostlib.fnnh_java_double(x, b)
        return b.value.decode()
    for x, s in [(1.0, "1.0"), (0.001, "0.001"), (1e-4, "1.0E-4"), (1e7, "1.0E7"), (9999999.999, "9999999.999"),
                 (123456.789, "123456.789"), (0.1 + 0.2, "0.30000000000000004"), (1e21, "1.0E21"), (1.234e-5, "1.234E-5"),
                 (100.0, "100.0"), (0.0, "0.0"), (-2.5, "-2.5"), (2.0 ** -10, "9.765625E-4"), (1 / 3, "0.3333333333333333"),
                 (float("nan"), "NaN"), (float("inf"), "Infinity"), (12345678.0, "1.2345678E7")]:
        assert jd(x) == s, (x, jd(x), s)
    from oracle import csw_oracle as W
    n = 7
    D = oracle.synth(n, 3)
    order, _, _ = oracle.run(D)
    w, _ = W.split_weights(D, order)
    names = b"".join((f"tax on{i + 1}".encode()).ljust(256, b"\0") for i in range(n))
    hostlib.fnnh_write_nexus.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_double)]
    p = str(tmp_path / "out.nex")
    ns = hostlib.fnnh_write_nexus(p.encode(), n, D.ctypes.data_as(C.POINTER(C.c_double)), names,
                                  order.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(C.POINTER(C.c_double)))
    assert ns == int((w > 1e-6).sum())
    txt = open(p).read().split("\n")
    assert txt[0] == "#nexus" and txt[1] == "" and txt[2] == "BEGIN Taxa;" and txt[3] == f"DIMENSIONS ntax={n};"
    assert txt[5] == "[1] 'tax on1'"
    i = txt.index("BEGIN Distances;")
    assert txt[i + 2] == "FORMAT labels=no diagonal triangle=both;" and txt[i + 3] == "MATRIX"
    row0 = txt[i + 4].split(" ")
    assert row0[0] == "" and row0[1] == "0.0" and float(row0[2]) == D[0, 1] and len(row0) == n + 1
    i = txt.index("BEGIN Splits;")
    assert txt[i + 1] == f"DIMENSIONS ntax={n} nsplits={ns};"
    assert txt[i + 4] == "CYCLE " + " ".join(str(int(t)) for t in order[1:]) + ";"
    first = txt[i + 6]
    assert first.startswith("[1, size=") and first.endswith(",") and " \t " in first
    # every split line lists a contiguous arc of the cycle that avoids ordering[n]
    pos = {int(t): k for k, t in enumerate(order[1:])}
    for line in txt[i + 6: i + 6 + ns]:
        taxa = [int(t) for t in line.split(" \t ")[2].rstrip(",").split()]
        ks = sorted(pos[t] for t in taxa)
        assert ks == list(range(ks[0], ks[0] + len(ks))) and int(order[n]) not in taxa
        assert taxa == sorted(taxa)
    assert txt[-3] == "END; [st_Assumptions]" and txt[-2] == "" and txt[-1] == ""


@pytest.mark.parametrize("n", [2, 5, 33, 300, 2100])
def test_nexus_document_parallel_writer_equals_the_list_route(hostlib, tmp_path, n):
    """printNexusFromWeights (what the CLI and fnnh_write_nexus run: bands of rows and groups of splits formatted by all host
    threads, member lists never materialised) writes byte for byte the document of splitsFromWeights +
    printNexusWithSplitsAndDistances (OutputPrinter.java:8-96 line by line) - weights on both sides of the 1e-6
    threshold, in plain and in E notation, a random cycle; n = 2100 spans several bands and the multi-row tile gather."""
    sig = [C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    hostlib.fnnh_write_nexus.argtypes = sig
    hostlib.fnnh_write_nexus_lists.argtypes = sig
    rng = np.random.default_rng(n)
    D = rng.random((n, n)) * 10.0 ** rng.integers(-5, 9, (n, n))
    D = D + D.T
    np.fill_diagonal(D, 0.0)
    order = np.concatenate([[0], rng.permutation(n) + 1]).astype(np.int32)
    N = n * (n - 1) // 2
    w = np.zeros(N)
    idx = rng.choice(N, min(N, int(2.4 * n)), replace=False)
    w[idx] = rng.random(len(idx)) * 10.0 ** rng.integers(-8, 4, len(idx))
    names = b"".join((f"taxon {i + 1}".encode()).ljust(256, b"\0") for i in range(n))
    args = (n, D.ctypes.data_as(C.POINTER(C.c_double)), names, order.ctypes.data_as(C.POINTER(C.c_int32)),
            w.ctypes.data_as(C.POINTER(C.c_double)))
    pa, pb = str(tmp_path / "a.nex"), str(tmp_path / "b.nex")
    na = hostlib.fnnh_write_nexus(pa.encode(), *args)
    nb = hostlib.fnnh_write_nexus_lists(pb.encode(), *args)
    assert na == nb == int((w > 1e-6).sum())
    a, b = open(pa, "rb").read(), open(pb, "rb").read()
    assert a == b, next((i, a[max(0, i - 40): i + 40], b[max(0, i - 40): i + 40]) for i in range(min(len(a), len(b))) if a[i] != b[i])


# ---- a whole-document known answer (tests/golden/kat5.nex) -------------------------------------------------------------
# The 5-taxon known-answer case of SURVEY.md App. B (tests/golden/kat5.json: small-integer distances, circular order
# [0, 1, 2, 3, 4, 5] hand-traced from NetMakerOriginal.java) continued through the rest of FastNN.main: the live path's
# non-negative least-squares problem (FastNN.java:401-454) has the feasible Chepoi-Fichet solution
#   x(i, j) = (d(i, j) + d(i+1, j+1) - d(i, j+1) - d(i+1, j)) / 2   (positions modulo n)
# whose entries are half-integers here: in the live index order (FastNN.java:405-419: (0,1), (0,2), ... (3,4), split k =
# taxa ordering[i+1 .. j]) x = 1, 3.5, 1.5, 1, 1, 1, 0, 1, 1.5, 1.5 - nine splits above the 1e-6 threshold (:455).  Every
# number of the document is therefore exact, Double.toString prints "1.0" / "3.5" / "10.0", and the document is derived by
# hand from OutputPrinter.java:8-96 line by line: `[k, size=min(|S|, n - |S|)] \t w \t  members,` (:70-85), the distance
# block as " " + value per entry (:34-47), CYCLE from ordering[1..n] (:56-61), the fixed st_Assumptions block (:87-96).
KAT5_LIVE_WEIGHTS = [1.0, 3.5, 1.5, 1.0, 1.0, 1.0, 0.0, 1.0, 1.5, 1.5]


def _kat5():
    import json
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "kat5.json")))
    n = k["n"]
    D = np.zeros((n, n))
    for i, row in enumerate(k["lower_triangle_rows"]):
        for j, v in enumerate(row):
            D[i, j] = D[j, i] = float(v)
    return k, D, open(os.path.join(ROOT, "tests", "golden", "kat5.nex"), "rb").read()


def test_kat5_weights_are_the_live_paths_optimum():
    """the fixture's weights are the unique optimum of the reference's dense problem (solver-independent: A x = d exactly, x >= 0)"""
    from oracle import csw_oracle as W
    _, D, _ = _kat5()
    A = W.live_design_matrix(5, np.arange(6, dtype=np.int32))
    assert (A @ np.array(KAT5_LIVE_WEIGHTS) == W.packed_distances(D)).all()


def test_kat5_nexus_document_byte_for_byte(hostlib, tmp_path):
    """N2: the host writer reproduces the hand-derived document of the 5-taxon known-answer case byte for byte."""
    k, D, want = _kat5()
    n = 5
    sig = [C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    hostlib.fnnh_write_nexus.argtypes = sig
    hostlib.fnnh_write_nexus_lists.argtypes = sig
    names = b"".join((f"t{i + 1}".encode()).ljust(256, b"\0") for i in range(n))
    order = np.arange(n + 1, dtype=np.int32)
    w = np.array(KAT5_LIVE_WEIGHTS)
    for fn in (hostlib.fnnh_write_nexus, hostlib.fnnh_write_nexus_lists):
        p = str(tmp_path / "kat5.nex")
        ns = fn(p.encode(), n, D.ctypes.data_as(C.POINTER(C.c_double)), names, order.ctypes.data_as(C.POINTER(C.c_int32)),
                w.ctypes.data_as(C.POINTER(C.c_double)))
        assert ns == 9
        got = open(p, "rb").read()
        assert got == want, next((i, got[max(0, i - 30): i + 30], want[max(0, i - 30): i + 30]) for i in range(min(len(got), len(want))) if got[i] != want[i])


@pytest.mark.gpu
def test_cli_kat5_whole_document(tmp_path):
    """The CLI end to end on the known-answer Phylip file: order, split weights on the GPU (closed form: exact here) and the
    document - stdout equals tests/golden/kat5.nex byte for byte (FastNN.java:398-491, OutputPrinter.java:8-96)."""
    k, D, want = _kat5()
    p = tmp_path / "kat5.phy"
    p.write_text(k["phylip"])
    exe = os.path.join(PKG, "bin", "fastnn")
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "Canonical", "-order"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout == "[0, 1, 2, 3, 4, 5]\n", r.stderr
    r = subprocess.run([exe, "-distFile", str(p), "-mode", "Canonical"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want
