#!/bin/bash
# K5 at C5 under rocprofv3: kernel stats, then SQ counters.  gpurun -- 'bash scripts/prof_k5.sh <tag> [bench_inbreed.py args]'
set -e
TAG=${1:-k5}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT/stats $OUT/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/bench_inbreed.py ${@:2} > $OUT/stats.txt 2> $OUT/stats.err
python3 $REPO/scripts/kernel_stats.py $OUT/stats | tee $OUT/kernel_stats.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/scripts/bench_inbreed.py ${@:2} > $OUT/pmc.txt 2> $OUT/pmc.err
python3 $REPO/scripts/pmc_sum.py $OUT/pmc k_inbreed | tee $OUT/pmc_sum.txt
