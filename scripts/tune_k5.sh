#!/bin/bash
# K5 launch-shape sweep: gpurun -- 'bash scripts/tune_k5.sh'
for b in 4 6 8 12 16 24 32; do
  echo "KGX_K5_BLOCKS_PER_CU=$b"
  KGX_K5_BLOCKS_PER_CU=$b python3 scripts/bench_inbreed.py 2>&1 | grep "Simple"
done
