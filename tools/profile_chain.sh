#!/bin/bash
# Event-chain evidence: rocprofv3 kernel trace of one whole run + the engine's event log -> per-event-kind table
# of kernel durations and inter-kernel gaps.  usage: tools/profile_chain.sh <tag> [n]   (writes under gpurun_out/)
set -o pipefail
TAG=${1:-chain}
N=${2:-32768}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_${TAG}
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_${TAG} -- python3 $GRAFT_REPO_ROOT/tools/trace_run.py $N /tmp/events_${TAG}_${N}.npz > $OUT/${TAG}_trace_run.log 2>&1 || { tail -5 $OUT/${TAG}_trace_run.log; exit 2; }
grep total= $OUT/${TAG}_trace_run.log
CSV=$(find /tmp/trace_${TAG} -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/chain_table.py $CSV /tmp/events_${TAG}_${N}.npz $OUT/${TAG}_chain_table_n${N}.json $OUT/${TAG}_chain_table_n${N}.md
