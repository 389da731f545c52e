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
javac -nowarn -d "$OUT" "$FASTNN_REF_DIR/NetNode.java" "$FASTNN_REF_DIR/NetMakerOriginal.java" "$FASTNN_REF_DIR/NeighborNetCanonical.java" "$HERE/GoldenDriver.java"
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
  java -Xmx14g -cp "$OUT" nnet.GoldenDriver 1 "$c" | tee -a "$LINES"
done
