run() { # name env...
  name=$1; shift
  env "$@" FNN_SW_LOG=1 timeout -k 10 200 python tests/tools/splits_perf.py $N > gpurun_out/r4_ab2_${name}_$N.log 2>&1 || return 1
  grep "^n=" gpurun_out/r4_ab2_${name}_$N.log | cut -c1-200 | sed "s/^/$name: /"
}
for N in 16384 32768; do
run A_r03ratio_revive0 FNN_SW_RATIO_POLICY=0 FNN_SW_REVIVE_MINF=0 || exit 1
run B_r03ratio_revive8k FNN_SW_RATIO_POLICY=0 || exit 1
run C_newratio_revive0 FNN_SW_REVIVE_MINF=0 || exit 1
run D_r03ratio_norevive FNN_SW_RATIO_POLICY=0 FNN_SW_REVIVE=0 || exit 1
done
