#!/bin/bash
# K5 launch-shape sweeps: gpurun -- 'bash scripts/tune_k5.sh'
for g in 4 8; do
  echo "KGX_K5_EVAL_GPL=$g"
  KGX_K5_EVAL_GPL=$g python3 scripts/bench_inbreed.py 10000 1000000 --all 2>&1 | grep "HallME\|Loglik"
done
