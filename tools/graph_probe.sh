#!/bin/bash
# Experiment: batches of events launched as hipGraphs (FNN_GRAPH=1) against plain stream launches - wall clock of whole runs and
# the traced per-kernel durations (rocprofv3 --kernel-trace --stats), 32768 taxa.  -> gpurun_out/<tag>_graph_*
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 tools/quick_perf.py 32768 32768 > $OUT/${TAG}_graph_off.log 2>&1 || exit 1
grep total= $OUT/${TAG}_graph_off.log | cut -c1-90
export FNN_GRAPH=1
timeout -k 10 120 python3 tools/quick_perf.py 32768 32768 > $OUT/${TAG}_graph_on.log 2>&1 || { tail -5 $OUT/${TAG}_graph_on.log; exit 2; }
grep total= $OUT/${TAG}_graph_on.log | cut -c1-90
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_graph_${TAG} -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 32768 > $OUT/${TAG}_graph_on_rocprof.log 2>&1 || exit 3
find /tmp/prof_graph_${TAG} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_graph_on_kernel_stats.csv \;
head -8 $OUT/${TAG}_graph_on_kernel_stats.csv | cut -c1-150
