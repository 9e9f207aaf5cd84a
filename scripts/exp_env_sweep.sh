#!/bin/bash
# Experiment: bench_inbreed.py under a list of environment settings ("K=V K2=V2" per argument).
# gpurun -- 'bash scripts/exp_env_sweep.sh "<env A>" "<env B>" ...'   (arguments after -- go to bench_inbreed.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
ARGS="10000 5000000 --all"
for E in "$@"; do
  echo "== $E" | tee -a gpurun_out/exp_env_sweep.log
  env $E timeout -k 10 200 python3 scripts/bench_inbreed.py $ARGS 2>&1 | grep -v synth | sed 's/(first call.*passes)//' | tee -a gpurun_out/exp_env_sweep.log
done
