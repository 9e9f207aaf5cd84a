#!/bin/bash
# rocprofv3 kernel stats for the secondary sweeps (K3 by-genome at C3, K5 inbreeding at C5, K7 iterative estimators at C5).  gpurun -- 'bash scripts/profile_aux.sh r01'
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_aux
rm -rf $OUT && mkdir -p $OUT/k3 $OUT/k5 $OUT/k7
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k3 -- python3 $REPO/scripts/bench_by_genome.py > $OUT/k3.txt 2> $OUT/k3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k5 -- python3 $REPO/scripts/bench_inbreed.py > $OUT/k5.txt 2> $OUT/k5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k7 -- python3 $REPO/scripts/bench_inbreed.py 10000 5000000 --only-iterative > $OUT/k7.txt 2> $OUT/k7.err
cat $OUT/k3.txt $OUT/k5.txt $OUT/k7.txt
