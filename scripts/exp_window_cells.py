"""One-launch iteration at 2504 genomes over selection sizes: a wave against a block per genome (KGX_K7_WAVE_GENOMES)."""
import sys, time, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls
capi.init(0)
G, L = 2512, 400_000
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
rng = np.random.default_rng(1)
for n_sel in (1000, 1500, 2000, 2048):
    index = np.sort(rng.choice(L, n_sel, replace=False)).astype(np.uint32)
    sub = np.ascontiguousarray(table[index])
    for algo in ("HallME", "Loglikelihood"):
        start = capi.reference_starts(algo, 4242, 2504)
        for w in ("1", "1000000"):
            os.environ["KGX_K7_WAVE_GENOMES"] = w
            for _ in range(3):
                m.inbreed(sub, algo, phased=True, locus_index=index, g0=0, g1=2504, start=start)
            t0 = time.perf_counter()
            for _ in range(10):
                m.inbreed(sub, algo, phased=True, locus_index=index, g0=0, g1=2504, start=start)
            print(f"{algo} {n_sel} loci x 2504, {'wave' if w == '1' else 'block'} per genome: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
