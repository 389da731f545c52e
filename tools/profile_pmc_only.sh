#!/bin/bash
# PMC passes over the scheduled screening launches only (tools/profile_round.sh step 3), for a source whose host side changed after the round's profile
set -o pipefail
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_screen<true, true>" --output-format csv -d /tmp/pmc_${TAG}_$C -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 32768 > $OUT/${TAG}_pmc_$C.log 2>&1 || exit 3
  find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv" -exec cp {} /tmp/pmc_${TAG}_$C.csv \;
  grep total= $OUT/${TAG}_pmc_$C.log | cut -c1-120
done
BYTES=$(grep -o "timed_screen_bytes=[0-9]*" $OUT/${TAG}_pmc_FETCH_SIZE.log | cut -d= -f2)
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${TAG}_FETCH_SIZE.csv /tmp/pmc_${TAG}_WRITE_SIZE.csv $BYTES $OUT/${TAG}_pmc_screen_windows_summary_n32768.json "k_screen<true, true>" | cut -c1-400
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY --kernel-include-regex "k_track" --output-format csv -d /tmp/pmc_${TAG}_track -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 32768 > $OUT/${TAG}_pmc_track.log 2>&1 || exit 4
find /tmp/pmc_${TAG}_track -name "*counter_collection.csv" -exec cp {} /tmp/pmc_${TAG}_track.csv \;
python3 $GRAFT_REPO_ROOT/tools/pmc_kernel_summary.py /tmp/pmc_${TAG}_track.csv k_track $OUT/${TAG}_pmc_k_track_summary_n32768.json | cut -c1-300
