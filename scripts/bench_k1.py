"""K1 alone: bench.py's aux.k1_flatten (GPU_ALLELE over the oracle's createVariantDB input).   python scripts/bench_k1.py"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from kgl_gene_amd import capi  # noqa: E402

capi.ensure_built()
capi.init(0)
sys.argv = sys.argv[:1]
print(json.dumps(bench.aux_k1_flatten(bench.parse_args(), capi), indent=1))
