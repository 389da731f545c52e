#!/bin/bash
# Round measurements on the GPU box (everything lands under gpurun_out/<tag>_*; copy what is to be kept to profiles/<round>/):
#   1. the bench line (python3 bench.py);
#   2. rocprofv3 --kernel-trace --stats of the same command (without the CPU baseline and the extra chain run);
#   3. PMC passes over the scheduled screening launches (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md
#      prescribes) -> HBM bytes per launch vs the algorithmic bytes;
#   4. the event-chain evidence: per-event-kind table of kernel durations and gaps (tools/profile_chain.sh) at 32768 and 4096,
#      and the in-kernel phase split of k_track / k_update (FNN_TICKS=1);
#   5. the split-weight solver: kernel stats of one solve at 4096 taxa.
# usage: tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_n32768.json 2> $OUT/${TAG}_bench.err || exit 1
tail -c 600 $OUT/${TAG}_bench_n32768.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-chain --no-config5 > $OUT/${TAG}_bench_under_rocprof_n32768.json 2> $OUT/${TAG}_rocprof.err || exit 2
find /tmp/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats_n32768.csv \;
find /tmp/prof_${TAG} -name "*domain_stats.csv" -exec cp {} $OUT/${TAG}_domain_stats_n32768.csv \;
head -12 $OUT/${TAG}_kernel_stats_n32768.csv | cut -c1-140
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_screen<true, true>" --output-format csv -d /tmp/pmc_${TAG}_$C -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 32768 > $OUT/${TAG}_pmc_$C.log 2>&1 || exit 3
  find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv" -exec cp {} /tmp/pmc_${TAG}_$C.csv \;
  grep total= $OUT/${TAG}_pmc_$C.log | cut -c1-120
done
BYTES=$(grep -o "timed_screen_bytes=[0-9]*" $OUT/${TAG}_pmc_FETCH_SIZE.log | cut -d= -f2)
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${TAG}_FETCH_SIZE.csv /tmp/pmc_${TAG}_WRITE_SIZE.csv $BYTES $OUT/${TAG}_pmc_screen_windows_summary_n32768.json "k_screen<true, true>" | cut -c1-400
cd $GRAFT_REPO_ROOT
bash tools/profile_chain.sh ${TAG} 32768 > $OUT/${TAG}_chain_32768.log 2>&1; grep -E "window" $OUT/${TAG}_chain_32768.log | cut -c1-260
bash tools/profile_chain.sh ${TAG} 4096 > $OUT/${TAG}_chain_4096.log 2>&1; grep -E "window" $OUT/${TAG}_chain_4096.log | cut -c1-260
FNN_TICKS=1 python3 tools/quick_perf.py 4096 16384 32768 > $OUT/${TAG}_ticks.log 2>&1; cut -c1-220 $OUT/${TAG}_ticks.log
cd /tmp
#   4b. one PMC pass over k_track, the kernel that dominates the run (SQ block: 8 slots per pass)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY --kernel-include-regex "k_track" --output-format csv -d /tmp/pmc_${TAG}_track -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 32768 > $OUT/${TAG}_pmc_track.log 2>&1 || exit 4
find /tmp/pmc_${TAG}_track -name "*counter_collection.csv" -exec cp {} /tmp/pmc_${TAG}_track.csv \;
python3 $GRAFT_REPO_ROOT/tools/pmc_kernel_summary.py /tmp/pmc_${TAG}_track.csv k_track $OUT/${TAG}_pmc_k_track_summary_n32768.json | cut -c1-600
#   5. the split-weight solver: kernel stats of one solve at 4096 taxa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sw_${TAG} -- python3 $GRAFT_REPO_ROOT/tests/tools/splits_perf.py 4096 > $OUT/${TAG}_splits_perf_n4096.log 2>&1
find /tmp/prof_sw_${TAG} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_splits_kernel_stats_n4096.csv \;
tail -1 $OUT/${TAG}_splits_perf_n4096.log; head -12 $OUT/${TAG}_splits_kernel_stats_n4096.csv | cut -c1-140
cd $GRAFT_REPO_ROOT
#   6. the Relaxed mode (one workgroup searching for mutual row minima): whole runs and the kernel statistics of one
python3 tools/relaxed_perf.py 4096 16384 > $OUT/${TAG}_relaxed_perf.log 2>&1; cat $OUT/${TAG}_relaxed_perf.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rl_${TAG} -- python3 $GRAFT_REPO_ROOT/tools/relaxed_perf.py 4096 > /dev/null 2>&1
find /tmp/prof_rl_${TAG} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_relaxed_kernel_stats_n4096.csv \;
head -6 $OUT/${TAG}_relaxed_kernel_stats_n4096.csv | cut -c1-140
