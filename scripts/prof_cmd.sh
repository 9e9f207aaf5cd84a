#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python command (run on the GPU box from the repo root): prof_cmd.sh <tag> <script> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 "$@" > $OUT/run.log 2>&1
grep -v "rocprofv3\|amdgpu.ids" $OUT/run.log | tail -30
