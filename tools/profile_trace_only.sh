#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command and the event-chain table at 32768 taxa (steps 2 and 4 of tools/profile_round.sh),
# for a source whose kernels changed after the round's full profile.   usage: tools/profile_trace_only.sh <tag>
set -o pipefail
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-chain --no-config5 > $OUT/${TAG}_bench_under_rocprof_n32768.json 2> $OUT/${TAG}_rocprof.err || exit 2
find /tmp/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats_n32768.csv \;
find /tmp/prof_${TAG} -name "*domain_stats.csv" -exec cp {} $OUT/${TAG}_domain_stats_n32768.csv \;
head -10 $OUT/${TAG}_kernel_stats_n32768.csv | cut -c1-140
cd $GRAFT_REPO_ROOT
bash tools/profile_chain.sh ${TAG} 32768 > $OUT/${TAG}_chain_32768.log 2>&1; grep -E "window" $OUT/${TAG}_chain_32768.log | cut -c1-260
