"""The table passes with a locus index (a window's sampled loci): every k-th locus of a C5-sized matrix.
python scripts/bench_inbreed_indexed.py [genomes] [loci] [step]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
step = int(sys.argv[3]) if len(sys.argv) > 3 else 2
capi.init(0)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
index = np.arange(0, L, step, dtype=np.uint32)
sub = np.ascontiguousarray(table[index])
for algo in ("Simple", "RitlandLocus", "HallME"):
    m.inbreed(sub, algo, phased=True, locus_index=index)
    t = time.perf_counter()
    res = m.inbreed(sub, algo, phased=True, locus_index=index)
    dt = time.perf_counter() - t
    print(f"{algo}: {len(index)} of {L} loci x {G} genomes: wall {dt*1e3:.1f} ms; frequency sweep kernel {capi.inbreed_last_kernel_ms():.2f} ms = "
          f"{G * len(index) / capi.inbreed_last_kernel_ms() / 1e9:.2f} TB/s of selected bytes  mean F {res['inbred_allele_sum'].mean():+.4f}", flush=True)
