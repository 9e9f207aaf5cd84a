#!/bin/bash
# timing experiments on the K5 evaluation kernel (results are wrong by construction): gpurun -- 'bash scripts/exp_k5.sh'
for f in "-DKGX_EXP_NOBUILD" "-DKGX_EXP_NOBUILD -DKGX_EXP_NOBARRIER"; do
  echo "== $f"
  KGX_HIPCC_FLAGS="$f" python3 -c "from kgl_gene_amd import build; build.build_kgx(force=True)" 2>/dev/null
  python3 scripts/bench_inbreed.py 10000 1000000 --all 2>&1 | grep "HallME\|Loglik"
done
python3 -c "from kgl_gene_amd import build; build.build_kgx(force=True)"
