"""kgx_genome_row_lists (variant-major 2-bit rows -> per genome the rows it carries) on a 1000-Genomes-sized population:
python scripts/bench_row_lists.py [genomes variants]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2504
V = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
capi.ensure_built()
capi.init(0)
pop = capi.Population(G, V)
pop.synth_biallelic(1111, 0, 0)
matrix = pop.sweep_bytes
begin = np.zeros(G + 1, dtype=np.uint64)
for _ in range(2):
    t0 = time.perf_counter()
    capi.check(capi.lib().kgx_genome_row_lists(pop._h, 0, G, None, capi.ptr(begin), None, 0))
    t_count = time.perf_counter() - t0
entries = int(begin[-1])
print(f"{G} genomes x {V} rows: {entries:.4g} carried cells ({entries / (G * V):.1%}); count pass + offsets {t_count * 1e3:.1f} ms "
      f"= {matrix / t_count / 1e12:.2f} TB/s of the matrix", flush=True)
t0 = time.perf_counter()
b, rows = pop.genome_row_lists()
t_all = time.perf_counter() - t0
print(f"sizing + fill + download of {rows.nbytes / 1e9:.2f} GB of row numbers: {t_all * 1e3:.0f} ms ({entries / t_all / 1e9:.2f} G entries/s)", flush=True)
g = G // 2
codes_ok = np.all(np.diff(rows[int(b[g]):int(b[g + 1])].astype(np.int64)) > 0)
print("ascending within a genome:", bool(codes_ok))
pop.close()
