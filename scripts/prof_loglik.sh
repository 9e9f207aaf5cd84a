#!/bin/bash
# rocprofv3 kernel stats of Loglikelihood calls on the moments: C5 and a mid-size call (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ll
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/c5 -o c5 --output-format csv -- python3 scripts/dev_loglik_moments.py ${1:-10000} ${2:-5000000} 3 > $OUT/c5.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
