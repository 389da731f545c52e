"""OPT-IN Java legs of bench.py's cpu_baseline (B1 = `-threads 1`, B2 = `-threads <host cores>`).

Runs only where an operator has supplied a JDK and FASTNN_REF_DIR (a checkout of the reference); the build image and
the GPU box have neither, and then this returns the string "Java baseline unavailable (...)".  The reference's
sources are compiled where they lie; nothing of them is copied into this repository."""
import json
import os
import shutil
import subprocess
import tempfile


def run_if_available(root, n=4096, seed=1):
    ref = os.environ.get("FASTNN_REF_DIR", "")
    if not (shutil.which("javac") and shutil.which("java")):
        return "Java baseline unavailable (no javac / java on PATH)"
    if not os.path.isfile(os.path.join(ref, "NeighborNetCanonical.java")):
        return "Java baseline unavailable (FASTNN_REF_DIR does not point at the reference's sources)"
    here = os.path.dirname(os.path.abspath(__file__))
    out = tempfile.mkdtemp(prefix="fastnn_java_")
    try:
        subprocess.check_call(["javac", "-nowarn", "-d", out] + [os.path.join(ref, f) for f in
                              ("NetNode.java", "NetMakerOriginal.java", "NeighborNetCanonical.java")] +
                              [os.path.join(here, "GoldenDriver.java")])
        cores = min(len(os.sched_getaffinity(0)), 16)
        res = {}
        for name, threads in (("B1_threads_1", 1), (f"B2_threads_{cores}", cores)):
            # (B2 is a reported timing only: the reference's pool branch is tie-racy and breaks above 23170 live
            #  nodes; the driver exits explicitly because the reference never shuts its pool down)
            line = subprocess.run(["java", "-Xmx8g", "-cp", out, "nnet.GoldenDriver", str(threads), f"{n}:uniform53:{seed}"],
                                  capture_output=True, text=True, timeout=3600).stdout.strip().splitlines()[-1]
            r = json.loads(line)
            res[name] = {"n_taxa": n, "seconds": r["seconds"], "order_sha256": r["order_sha256"]}
        return res
    finally:
        shutil.rmtree(out, ignore_errors=True)
