set -e
for flag in "" "-DKGX_EXP_NOMATH" "-DKGX_EXP_NOLOAD"; do
  KGX_HIPCC_FLAGS="$flag" python -m kgl_gene_amd.build > /dev/null 2>&1
  echo "== flags: '$flag'"
  python scripts/bench_inbreed.py 10000 5000000 --only-iterative 2>&1 | grep -E "HallME|Loglik"
done
python -m kgl_gene_amd.build --force > /dev/null 2>&1
