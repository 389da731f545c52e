#!/bin/bash
# Development aid: the share of departed splits in the factor at which the split-weight solver rebuilds it (FNN_SW_RFRAC), several seeds
# on ONE box in one call (the solver's time moves by +-10 % with any rule that alters its path; a rebuild does not alter it).
# usage: tools/sweep_sw_rfrac.sh "<seeds>" "<shares>" [n]
SEEDS=${1:-"1 2"}; SHARES=${2:-"0.15 0.23 0.30"}; N=${3:-32768}
for s in $SEEDS; do for f in $SHARES; do
  FNN_SW_RFRAC=$f timeout -k 10 300 python tests/tools/e2e_run.py $N $s --no-nexus --no-kkt --json gpurun_out/r04_rfrac_n${N}_seed${s}_$f.json > gpurun_out/r04_rfrac_n${N}_seed${s}_$f.log 2>&1 || exit 1
  python - gpurun_out/r04_rfrac_n${N}_seed${s}_$f.json $s $f <<'PY'
import json, sys
w = json.load(open(sys.argv[1]))["weights"]
print(f"seed {sys.argv[2]} share {sys.argv[3]}: {w['t_solve_s'] - w['t_alloc_s']:.2f} s net of hipMalloc, steps {w['outer_iterations']} rebuilds {w['refactorizations']} solves {w['solves']} "
      f"entered {w['entered']} departed {w['departed']} splits {w['nsplits']} kkt {w['kkt_violation']:.2e}", flush=True)
PY
done; done
