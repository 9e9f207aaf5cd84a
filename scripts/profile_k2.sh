#!/bin/bash
# rocprofv3 evidence for the bench's dominant kernel.  Run on the GPU box via gpurun:
#   gpurun -- 'bash scripts/profile_k2.sh r01'
# Kernel trace/stats and the two PMC passes are separate runs (gpurun refuses mixed modes; FETCH_SIZE
# and WRITE_SIZE do not fit one pass on gfx950).
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -type f | head -50
du -sh $OUT
