#!/bin/bash
# LDS / VALU counters of the table passes at C5: gpurun -- 'bash scripts/pmc_lds.sh <tag> [bench_inbreed.py args]'
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_lds_$1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -- python3 $REPO/scripts/bench_inbreed.py ${@:2} > $OUT/out.txt 2> $OUT/err.txt
python3 $REPO/scripts/pmc_sum.py $OUT k_inbreed_eval | tee $OUT/pmc_sum.txt
tail -4 $OUT/out.txt
