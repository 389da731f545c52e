#!/bin/bash
# OPT-IN: true goldens from the reference itself.  Needs a JDK (javac, java) and FASTNN_REF_DIR pointing at a checkout
# of JacobPorter/FastNeighborNet; neither exists in the build image or on the GPU box, so this is for an operator's
# machine.  Compiles the reference's three pure-JDK files where they lie (nothing is copied into this repository)
# together with GoldenDriver.java and writes tests/golden/java_orders.json, which tests/test_java_golden.py consumes:
# with it the oracle is PINNED (small cases are re-run by the oracle, the BASELINE sizes are compared by hash with
# tests/golden/oracle_big.json, which the GPU engine reproduces event by event).
#
#   FASTNN_REF_DIR=/path/to/FastNeighborNet tests/golden/java/make_java_golden.sh            # all default cases
#   FASTNN_REF_DIR=... tests/golden/java/make_java_golden.sh 4096:uniform53:1 ...           # chosen cases
#
# The default list ends with the BASELINE sizes 16384 and 32768 (hashes only; the reference's single-threaded scan needs
# roughly 1-2 h and 8-16 h for them): every finished case is kept - interrupt the run when the small ones are enough.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
command -v javac >/dev/null 2>&1 && command -v java >/dev/null 2>&1 || { echo "Java baseline unavailable: no javac/java on PATH"; exit 3; }
[ -n "${FASTNN_REF_DIR:-}" ] && [ -f "$FASTNN_REF_DIR/NeighborNetCanonical.java" ] || { echo "Java baseline unavailable: set FASTNN_REF_DIR to the reference checkout"; exit 3; }
OUT="$(mktemp -d)"
ROOT="$(cd "$HERE/../../.." && pwd)"
# the Java side of the drop-in (jni/NeighborNetHIP.java extends the reference's NetMakerOriginal) compiles with the reference
javac -nowarn -d "$OUT" "$FASTNN_REF_DIR/NetNode.java" "$FASTNN_REF_DIR/NetMakerOriginal.java" "$FASTNN_REF_DIR/NeighborNetCanonical.java" \
      "$ROOT/jni/NeighborNetHIP.java" "$HERE/GoldenDriver.java"
# ... and, where the engine's library, jni.h and a GPU exist, it RUNS: the JNI shim is built and every case also goes
# through NeighborNetHIP -> fastnn_jni.c -> libfastnn_hip.so ("hip_order_sha256" beside the reference's "order_sha256")
ENGINE=()
JH="${JAVA_HOME:-$(dirname "$(dirname "$(readlink -f "$(command -v javac)")")")}"
if [ -f "$ROOT/fastneighbornet_amd/libfastnn_hip.so" ] && [ -f "$JH/include/jni.h" ] && [ "${FASTNN_JNI:-1}" != "0" ]; then
  gcc -shared -fPIC -O2 -I"$JH/include" -I"$JH/include/linux" -I"$ROOT/include" "$ROOT/jni/fastnn_jni.c" \
      -L"$ROOT/fastneighbornet_amd" -lfastnn_hip -Wl,-rpath,"$ROOT/fastneighbornet_amd" -o "$OUT/libfastnn_jni.so"
  if python3 -c "import sys; sys.path.insert(0, '$ROOT'); import fastneighbornet_amd as fa; sys.exit(0 if fa.api().device_count() > 0 else 1)" 2>/dev/null; then
    ENGINE=(hip)
    echo "JNI drop-in built; a HIP device is visible: every case also runs through nnet.NeighborNetHIP"
  else
    echo "JNI drop-in built ($OUT/libfastnn_jni.so) but no HIP device is visible: reference legs only"
  fi
else
  echo "JNI drop-in not built (needs fastneighbornet_amd/libfastnn_hip.so and \$JAVA_HOME/include/jni.h): reference legs only"
fi
CASES=("$@")
[ ${#CASES[@]} -gt 0 ] || CASES=(4:uniform53:1 5:uniform53:1 9:dec4:2 17:uniform53:2 64:uniform53:1 64:dec4:1 200:uniform53:1 200:dec4:2 1030:dec4:4
                                 2048:uniform53:7 4096:uniform53:1 4096:dec4:1 4096:uniform53:2 4096:uniform53:3
                                 16384:uniform53:1 16384:uniform53:2 16384:uniform53:3 32768:uniform53:1)
LINES="$OUT/cases.jsonl"
: > "$LINES"
finish() {
  {
    echo '{"generator": "the reference (NeighborNetCanonical, -threads 1 semantics) via tests/golden/java/make_java_golden.sh", "java": "'"$(java -version 2>&1 | head -1 | tr -d '"')"'", "cases": ['
    paste -sd, "$LINES"
    echo ']}'
  } > "$HERE/../java_orders.json"
  echo "wrote $HERE/../java_orders.json ($(wc -l < "$LINES") cases)"
  rm -rf "$OUT"
}
trap finish EXIT
for c in "${CASES[@]}"; do
  java -Xmx14g -Djava.library.path="$OUT" -cp "$OUT" nnet.GoldenDriver ${ENGINE[@]+"${ENGINE[@]}"} 1 "$c" | tee -a "$LINES"
done
