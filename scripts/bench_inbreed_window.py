"""Window-sized kgx_inbreed calls (what the INBREED package issues: ~1000 sampled loci x one super population's genomes)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
G, L = 2512, 200_000
n_sel = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g0, g1 = (0, int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)        # [genomes: the first N] (default: 512 in the middle)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
index = np.sort(np.random.default_rng(1).choice(L, n_sel, replace=False)).astype(np.uint32)
sub = np.ascontiguousarray(table[index])
for _ in range(200):                      # clocks up
    m.inbreed(sub, "Simple", phased=True, locus_index=index, g0=g0, g1=g1)
for algo in ("Simple", "RitlandLocus", "HallME", "Loglikelihood"):
    m.inbreed(sub, algo, phased=True, locus_index=index, g0=g0, g1=g1)
    t0 = time.perf_counter()
    reps = 100
    for _ in range(reps):
        res = m.inbreed(sub, algo, phased=True, locus_index=index, g0=g0, g1=g1)
    dt = (time.perf_counter() - t0) / reps
    extra = f", {capi.inbreed_last_evaluations()} evaluations" if algo == "Loglikelihood" else ""
    print(f"{algo}: {dt*1e3:.3f} ms per call ({n_sel} loci x {g1-g0} genomes{extra})  mean F {res['inbred_allele_sum'].mean():+.4f}", flush=True)
