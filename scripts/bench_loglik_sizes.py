"""Loglikelihood between the window-sized calls and C5: ms per call over selection sizes (seeded reference starts)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
G, L = 2512, 400_000
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
rng = np.random.default_rng(1)
for n_sel in (2000, 8000, 9000, 30_000, 100_000, 400_000):
    index = np.sort(rng.choice(L, n_sel, replace=False)).astype(np.uint32)
    sub = np.ascontiguousarray(table[index])
    for g1 in (512, 2504):
        start = capi.reference_starts("Loglikelihood", 4242, g1)
        for _ in range(2):
            m.inbreed(sub, "Loglikelihood", phased=True, locus_index=index, g0=0, g1=g1, start=start)
        t0 = time.perf_counter()
        for _ in range(5):
            m.inbreed(sub, "Loglikelihood", phased=True, locus_index=index, g0=0, g1=g1, start=start)
        print(f"{n_sel:7d} loci x {g1:5d} genomes: {(time.perf_counter() - t0) / 5 * 1e3:8.3f} ms per call, {capi.inbreed_last_evaluations()} evaluations", flush=True)
