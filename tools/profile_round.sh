#!/bin/bash
# Round measurements on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) over the scheduled screening
# launches.  usage: tools/profile_round.sh <tag>   (writes under gpurun_out/)
set -o pipefail
TAG=${1:-la}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 1 > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err || exit 1
tail -c 1500 $OUT/bench_${TAG}.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_${TAG}_prof.json 2> $OUT/rocprof_${TAG}.err || exit 2
find /tmp/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
find /tmp/prof_${TAG} -name "*domain_stats.csv" -exec cp {} $OUT/${TAG}_domain_stats.csv \;
head -14 $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_screen<true, true>" --output-format csv -d /tmp/pmc_${TAG}_$C -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 32768 > $OUT/pmc_${TAG}_$C.log 2>&1 || exit 3
  find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv" -exec cp {} $OUT/pmc_${TAG}_$C.csv \;
  grep total= $OUT/pmc_${TAG}_$C.log | cut -c1-120
done
BYTES=$(grep -o "timed_screen_bytes=[0-9]*" $OUT/pmc_${TAG}_FETCH_SIZE.log | cut -d= -f2)
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_${TAG}_FETCH_SIZE.csv $OUT/pmc_${TAG}_WRITE_SIZE.csv $BYTES $OUT/pmc_${TAG}_summary.json "k_screen<true, true>"
