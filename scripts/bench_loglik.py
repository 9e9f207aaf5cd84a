"""Loglikelihood at C5 (10 k genomes x 5 M loci): the per-genome moments + the exact walk next to the floor (kgx_kernels_loglik.h)
against the passes over the bytes (two evaluations each).  --moments-only: the moments alone (for the counter passes)."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls (the library reads them at kgx_init / kgx_reload_options)
capi.init(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
G, L = (int(args[0]), int(args[1])) if len(args) > 1 else (10_000, 5_000_000)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
start = capi.reference_starts("Loglikelihood", 4242, G)
results = {}
variants = (("moments", None),) if "--moments-only" in sys.argv else (("moments", None), ("passes", "1"))
for label, env in variants:
    if env:
        os.environ["KGX_K7_LL_PASSES"] = env
    else:
        os.environ.pop("KGX_K7_LL_PASSES", None)
    m.inbreed(table, "Loglikelihood", phased=True, start=start)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = m.inbreed(table, "Loglikelihood", phased=True, start=start)
        walls.append(time.perf_counter() - t0)
    results[label] = res["inbred_allele_sum"].copy()
    print(f"Loglikelihood {label}: {np.median(walls) * 1e3:.1f} ms per call ({G} genomes x {L} loci), path '{capi.inbreed_last_path()}', "
          f"{capi.inbreed_last_evaluations()} evaluations, frequency sweep {capi.inbreed_last_sweep_ms():.2f} ms, class passes {capi.inbreed_last_moments_ms():.2f} ms, "
          f"search {capi.inbreed_last_search_ms():.2f} ms; mean F {results[label].mean():+.6f}", flush=True)
if len(results) == 2:
    d = np.abs(results["moments"] - results["passes"])
    print(f"|dF| max {d.max():.3e}, {int((d == 0).sum())} of {G} genomes to the bit", flush=True)
