"""ms per kgx_inbreed call over selection sizes around the one-launch iteration's switches (k_inbreed_iterate_genome:
a wave or a block per genome, 2..32 cells per thread; past 8192 loci the multi-kernel paths), both iterative estimators,
512 and 2504 genomes.  usage: bench_window_sizes.py [KGX_K7_WAVE_LOCI values ...]  (default: the library's default)"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls

capi.init(0)
G, L = 2512, 40_000
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
rng = np.random.default_rng(1)
sizes = (500, 1000, 1024, 1025, 1500, 2000, 2048, 2049, 3000, 4096, 8192, 8193, 12000, 20000)
settings = sys.argv[1:] or [""]
print(f"{'loci':>6} {'genomes':>7} {'algorithm':>14} " + " ".join(f"{('WAVE_LOCI=' + s) if s else 'default':>16}" for s in settings))
for n in (512, 2504):
    for n_sel in sizes:
        index = np.sort(rng.choice(L, n_sel, replace=False)).astype(np.uint32)
        sub = np.ascontiguousarray(table[index])
        for algo in ("HallME", "Loglikelihood"):
            start = capi.reference_starts(algo, 4242, n)
            cells = []
            for s in settings:
                if s:
                    os.environ["KGX_K7_WAVE_LOCI"] = s
                else:
                    os.environ.pop("KGX_K7_WAVE_LOCI", None)
                if hasattr(capi, "reload_options"):
                    capi.reload_options()
                for _ in range(3):
                    m.inbreed(sub, algo, phased=True, locus_index=index, g0=0, g1=n, start=start)
                reps = 20 if n_sel <= 8192 else 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    m.inbreed(sub, algo, phased=True, locus_index=index, g0=0, g1=n, start=start)
                cells.append((time.perf_counter() - t0) / reps * 1e3)
            print(f"{n_sel:>6} {n:>7} {algo:>14} " + " ".join(f"{c:>13.3f} ms" for c in cells), flush=True)
m.close()
