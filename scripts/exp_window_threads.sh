#!/bin/bash
# Window-sized calls: a block per genome against a wave per genome (KGX_K7_WAVE_GENOMES) over genome counts.
for g in 256 512 1024 1536 2504; do
  for w in 1 1000000; do
    echo "== genomes $g, KGX_K7_WAVE_GENOMES=$w"
    KGX_K7_WAVE_GENOMES=$w python scripts/bench_inbreed_window.py 1000 $g 2>/dev/null | grep "HallME\|Loglik"
  done
done
