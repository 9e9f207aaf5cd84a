#!/bin/bash
# PMC counters for the K5 kernels at C5: gpurun -- 'bash scripts/pmc_k5.sh "<counters>" <tag> [bench_inbreed.py args]'
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_k5_$2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/scripts/bench_inbreed.py ${@:3} > $OUT/out.txt 2> $OUT/err.txt
tail -2 $OUT/out.txt
