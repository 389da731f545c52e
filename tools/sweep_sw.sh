#!/bin/bash
cd $GRAFT_REPO_ROOT
for kf in 0.05 0.1 0.2; do for rf in 0.15 0.3; do
  echo "== kfrac $kf rfrac $rf"
  FNN_SW_KFRAC=$kf FNN_SW_RFRAC=$rf FNN_SW_LOG=1 timeout -k 10 200 python tests/tools/splits_perf.py 16384 2>&1 | grep "done:\|^n=\|GEMM work"
done; done
