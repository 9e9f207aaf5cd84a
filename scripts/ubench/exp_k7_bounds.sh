# The three-build experiment of profiles/r03_power_cap.md: the product, the pass without its fp64 work, the pass without its HBM
# reads.  The two experiment branches are NOT in the product's kernels: apply scripts/ubench/power_cap_experiment.patch first
# (git apply scripts/ubench/power_cap_experiment.patch), run this from the repo root, and revert it afterwards.
set -e
for flag in "" "-DKGX_EXP_NOMATH" "-DKGX_EXP_NOLOAD"; do
  KGX_HIPCC_FLAGS="$flag" python -m kgl_gene_amd.build > /dev/null 2>&1
  echo "== flags: '$flag'"
  python scripts/bench_inbreed.py 10000 5000000 --only-iterative 2>&1 | grep -E "HallME|Loglik"
done
python -m kgl_gene_amd.build --force > /dev/null 2>&1
