"""Loglikelihood passes on a population with realistic inbreeding (F in [0, 0.1]) vs the +-0.5 grid of the benchmarks."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
G, L = 10_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
table = np.full((L, 2), np.nan)
table[:, 0] = rng.uniform(0.02, 0.5, L).astype(np.float32)
for name, F in (("F in [0, 0.1]", rng.uniform(0.0, 0.1, G)), ("F in [-0.1, 0.1]", rng.uniform(-0.1, 0.1, G)), ("F on the -0.5 .. 0.5 grid", (np.arange(G) % 101) * 0.01 - 0.5)):
    m = capi.GenotypeMatrix(G, L)
    m.synth_inbred(table, F, seed=5)
    m.inbreed(table, "Loglikelihood", phased=True)
    t0 = time.perf_counter(); res = m.inbreed(table, "Loglikelihood", phased=True); dt = time.perf_counter() - t0
    err = np.abs(res["inbred_allele_sum"] - F)
    print(f"{name}: {capi.inbreed_last_evaluations()} evaluations, {dt*1e3:.0f} ms; |F_hat - F| median {np.median(err):.4f} max {err.max():.4f}", flush=True)
    m.close()
