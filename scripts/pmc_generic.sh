#!/bin/bash
# rocprofv3 --pmc over scripts/bench_inbreed.py: gpurun -- 'bash scripts/pmc_generic.sh <tag> "<counters>" <needle> [bench_inbreed.py args]'
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/scripts/bench_inbreed.py ${@:4} > $OUT/out.txt 2> $OUT/err.txt
python3 $REPO/scripts/pmc_sum.py $OUT $3 | tee $OUT/pmc_sum.txt
python3 $REPO/scripts/kernel_stats.py $OUT | head -8
