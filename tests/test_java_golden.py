"""True goldens from the reference itself - when an operator has generated them.

tests/golden/java/make_java_golden.sh (needs a JDK and FASTNN_REF_DIR; neither exists in the build image) runs the
reference's `new NeighborNetCanonical(D, n, 1, null).runNeighborNet()` on the synthetic matrices and writes
tests/golden/java_orders.json.  With that file present the oracle is PINNED: its order must equal the reference's for
every case.  Without it this test says so: PARITY UNPINNED."""
import hashlib
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_against_java_goldens(oracle):
    path = os.path.join(GOLD, "java_orders.json")
    if not os.path.exists(path):
        pytest.skip("PARITY UNPINNED: no Java goldens (tests/golden/java/make_java_golden.sh needs a JDK and FASTNN_REF_DIR)")
    doc = json.load(open(path))
    big = {}
    if os.path.exists(os.path.join(GOLD, "oracle_big.json")):
        big = {(c["n"], c["dist"], c["seed"]): c["order_sha256"] for c in json.load(open(os.path.join(GOLD, "oracle_big.json")))["cases"]}
    checked = 0
    for c in doc["cases"]:
        key = (c["n"], c["dist"], c["seed"])
        if "hip_order_sha256" in c:  # the case also went through jni/NeighborNetHIP.java -> libfastnn_hip.so on the operator's GPU
            assert c["hip_order_sha256"] == c["order_sha256"], ("the JNI drop-in disagrees with the reference", key)
        if c["n"] <= 2100:
            order, _, _ = oracle.run(oracle.synth(c["n"], c["seed"], c["dist"]), threads=1, want_events=False)
            assert hashlib.sha256(order.tobytes()).hexdigest() == c["order_sha256"], key
            if c.get("order"):
                assert order.tolist() == c["order"], key
            checked += 1
        elif key in big:
            assert big[key] == c["order_sha256"], key
            checked += 1
    assert checked > 0


def test_java_harness_reports_unavailable_without_a_jdk():
    """The bench's Java legs degrade to a message, never to an error, where no JDK exists."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("java_baseline", os.path.join(GOLD, "java", "java_baseline.py"))
    jb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(jb)
    if shutil.which("javac") and os.environ.get("FASTNN_REF_DIR"):
        pytest.skip("a JDK and FASTNN_REF_DIR are present: the real legs run in bench.py")
    assert str(jb.run_if_available(os.path.dirname(GOLD))).startswith("Java baseline unavailable")
