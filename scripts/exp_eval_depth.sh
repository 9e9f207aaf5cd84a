#!/bin/bash
# Experiment: the table passes at other read-ahead depths (rebuilds libkgx with KGX_HIPCC_FLAGS on the GPU box).
# gpurun -- 'bash scripts/exp_eval_depth.sh "<flags A>" "<flags B>" ...'
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
for FLAGS in "$@"; do
  echo "== $FLAGS" | tee -a gpurun_out/exp_eval_depth.log
  KGX_HIPCC_FLAGS="$FLAGS" python3 -c "from kgl_gene_amd import build; build.build_kgx(force=True)" > /dev/null 2>&1
  timeout -k 10 200 python3 scripts/bench_inbreed.py 10000 5000000 --all 2>&1 | grep -v synth | tee -a gpurun_out/exp_eval_depth.log
done
