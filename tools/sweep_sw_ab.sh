#!/bin/bash
# Development aid: policy variants of the split-weight solver on ONE box in ONE call (the method's path is sensitive to its rules and
# boxes differ by a few per cent, so variants are only comparable this way).  profiles/r04/r04_splits_policy_ab.md holds round 4's table
# (its variant C - the ratio step only for the retry of a block - was removed from the code after that measurement); for the rule that
# shipped - departed splits return only when no candidate is left, FNN_SW_REVIVE=2 - the comparison was repeated on three seeds with
# tests/tools/e2e_run.py (same file).
run() { # name env...
  name=$1; shift
  env "$@" FNN_SW_LOG=1 timeout -k 10 200 python tests/tools/splits_perf.py $N > gpurun_out/sw_ab_${name}_$N.log 2>&1 || return 1
  grep "^n=" gpurun_out/sw_ab_${name}_$N.log | cut -c1-200 | sed "s/^/$name: /"
}
for N in 16384 32768; do
run A_shipped || exit 1
run E_returns_at_every_step FNN_SW_REVIVE=1 || exit 1
run D_no_returns FNN_SW_REVIVE=0 || exit 1
done
