"""Cost of ONE objective evaluation by the moments for every genome at a fixed F (under rocprofv3: the k_loglik_search launches
in order: each F twice).  usage: dev_loglik_eval_cost.py genomes loci F1,F2,..."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from kgl_gene_amd import capi

G, L = int(sys.argv[1]), int(sys.argv[2])
points = [float(x) for x in sys.argv[3].split(",")]
capi.init(0)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
for F in points:
    for rep in range(2):
        t0 = time.perf_counter()
        v = m.inbreed_objective(table, np.full(G, F), phased=True)
        print(f"F = {F:+.3f}: {(time.perf_counter() - t0) * 1e3:.2f} ms per call, mean value {np.nanmean(v):.3f}, {int(np.isnan(v).sum())} handed over (first {np.flatnonzero(np.isnan(v))[:5]})", flush=True)
m.close()
