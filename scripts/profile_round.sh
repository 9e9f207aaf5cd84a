#!/bin/bash
# rocprofv3 evidence for one round: the default bench.py run (K2 at C3 + the aux sweeps K3 at C3 and K5 at C5) under
# --kernel-trace --stats, then one FETCH_SIZE and one WRITE_SIZE pass of the same command (separate runs: gpurun refuses
# mixed modes, and the two counters do not fit one pass on gfx950), then the K5/K7 table passes with SQ counters.
#   gpurun -- 'bash scripts/profile_round.sh r02'   then   python scripts/summarize_round.py r02
set -e
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done" > $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
echo "write done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k7_trace -- python3 $REPO/scripts/bench_inbreed.py 10000 5000000 --all > $OUT/k7.txt 2> $OUT/k7.err
echo "k7 trace done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/k7_sq -- python3 $REPO/scripts/bench_inbreed.py 10000 5000000 --all > $OUT/k7_sq.txt 2> $OUT/k7_sq.err
echo "k7 sq done" >> $OUT/progress.txt
# the regime the INBREED package runs in: window-sized calls, the whole iteration inside k_inbreed_iterate_genome
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/window_trace -- python3 $REPO/scripts/bench_inbreed_window.py 1000 > $OUT/window.txt 2> $OUT/window.err
echo "window trace done" >> $OUT/progress.txt
cat $OUT/k7.txt $OUT/window.txt
du -sh $OUT
# HallME over a large call on per-genome moments (kgx_kernels_hall.h) against the 50 passes, C5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hall_trace -- python3 $REPO/scripts/bench_hall.py > $OUT/hall.txt 2> $OUT/hall.err
echo "hall trace done" >> $OUT/progress.txt
cat $OUT/hall.txt
# Loglikelihood over a large call on the same moments (kgx_kernels_loglik.h) against the passes, C5; then the moments alone under the
# traffic and SQ counters (the pass that leaves the classes' hits as bit rows, the matrix-core moment passes, the search)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loglik_trace -- python3 $REPO/scripts/bench_loglik.py > $OUT/loglik.txt 2> $OUT/loglik.err
echo "loglik trace done" >> $OUT/progress.txt
cat $OUT/loglik.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/loglik_fetch -- python3 $REPO/scripts/bench_loglik.py --moments-only > $OUT/loglik_fetch.txt 2> $OUT/loglik_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/loglik_write -- python3 $REPO/scripts/bench_loglik.py --moments-only > $OUT/loglik_write.txt 2> $OUT/loglik_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/loglik_sq -- python3 $REPO/scripts/bench_loglik.py --moments-only > $OUT/loglik_sq.txt 2> $OUT/loglik_sq.err
echo "loglik counters done" >> $OUT/progress.txt
# the batched window regime: one kgx_inbreed_batch of 16 windows x 2504 genomes per launch pair
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/batch_trace -- python3 $REPO/scripts/bench_inbreed_batch.py > $OUT/batch.txt 2> $OUT/batch.err
echo "batch trace done" >> $OUT/progress.txt
cat $OUT/batch.txt
