"""One HallME and one Loglikelihood call just past the one-launch size (under rocprofv3: the kernels of the last call in order)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from kgl_gene_amd import capi
G, L = 2504, 12000
capi.init(0)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
alg = sys.argv[1] if len(sys.argv) > 1 else "Loglikelihood"
start = capi.reference_starts(alg, 7, G)
import time
for i in range(6):
    t = time.perf_counter(); m.inbreed(table, alg, phased=True, start=start); dt = time.perf_counter() - t
print(alg, "last call", round(dt * 1e3, 3), "ms, path", capi.inbreed_last_path())
