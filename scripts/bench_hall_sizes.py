"""HallME between the window-sized calls and C5: per-genome moments against the 50 passes over selection sizes."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls (the library reads them at kgx_init / kgx_reload_options)

capi.init(0)
G, L = 2512, 400_000
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
rng = np.random.default_rng(1)
for n_sel in (3000, 10_000, 30_000, 100_000, 400_000):
    index = np.sort(rng.choice(L, n_sel, replace=False)).astype(np.uint32)
    sub = np.ascontiguousarray(table[index])
    for g1 in (512, 2504):
        start = capi.reference_starts("HallME", 4242, g1)
        line = []
        for label, env in (("moments", None), ("passes", "1")):
            if env:
                os.environ["KGX_K7_HALL_PASSES"] = env
            else:
                os.environ.pop("KGX_K7_HALL_PASSES", None)
            for _ in range(3):
                m.inbreed(sub, "HallME", phased=True, locus_index=index, g0=0, g1=g1, start=start)
            t0 = time.perf_counter()
            for _ in range(10):
                m.inbreed(sub, "HallME", phased=True, locus_index=index, g0=0, g1=g1, start=start)
            line.append(f"{label} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms")
        print(f"{n_sel:7d} loci x {g1:5d} genomes: " + "   ".join(line), flush=True)
