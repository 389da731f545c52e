#!/bin/bash
# Diagnostic: what do the GPU's clocks, power and activity read while the latency-bound event chain runs?
# (rocm-smi is read-only here; sampled every ~0.25 s beside three 32768-taxon runs)  -> gpurun_out/<tag>_clock_probe.log
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_clock_probe.log
python3 tools/quick_perf.py 32768 32768 32768 32768 > gpurun_out/${TAG}_clock_probe_run.log 2>&1 &
PID=$!
: > $OUT
for i in $(seq 1 200); do
  if ! kill -0 $PID 2>/dev/null; then break; fi
  echo "--- sample $i $(date +%s.%N)" >> $OUT
  rocm-smi --showclocks --showuse --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|GPU use|Power" >> $OUT
  sleep 0.25
done
wait $PID
grep -E "total=" gpurun_out/${TAG}_clock_probe_run.log | cut -c1-100
grep -E "sclk" $OUT | sort | uniq -c | sort -rn | head -8
grep -E "Power" $OUT | sort | uniq -c | sort -rn | head -5
grep -E "GPU use" $OUT | sort | uniq -c | sort -rn | head -5
