"""Loglikelihood on the moments against the passes: objective values at random points, the search's results, the time per call.
usage: dev_loglik_moments.py [genomes] [loci] [reps]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls

G = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
capi.init(0)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
rng = np.random.default_rng(3)
if G * L <= 4_000_000_000:
    for trial in range(3):
        at = rng.uniform(-1.0, 1.0, G) if trial else np.linspace(-1.0, 1.0, G)
        t0 = time.perf_counter()
        a = m.inbreed_objective(table, at, phased=True)
        t1 = time.perf_counter()
        b = m.inbreed_objective(table, at, phased=True, by_passes=True)
        t2 = time.perf_counter()
        ok = np.isfinite(a)
        rel = np.abs(a[ok] - b[ok]) / np.abs(b[ok])
        worst = int(np.argmax(rel))
        print(f"objective trial {trial}: {ok.sum()} of {G} by moments ({(t1 - t0) * 1e3:.1f} ms; pass {(t2 - t1) * 1e3:.1f} ms), "
              f"|rel err| max {rel.max():.2e} at F = {at[ok][worst]:+.4f} (values {a[ok][worst]:.6f} / {b[ok][worst]:.6f}), median {np.median(rel):.1e}", flush=True)
start = capi.reference_starts("Loglikelihood", 4242, G)
results = {}
for label, env in (("moments", {}), ("passes", {"KGX_K7_LL_PASSES": "1"})):
    for k, v in env.items():
        os.environ[k] = v
    times = []
    for i in range(reps):
        t0 = time.perf_counter()
        r = m.inbreed(table, "Loglikelihood", phased=True, start=start)
        times.append((time.perf_counter() - t0) * 1e3)
    for k in env:
        del os.environ[k]
    results[label] = r["inbred_allele_sum"].copy()
    print(f"{label}: path '{capi.inbreed_last_path()}', {capi.inbreed_last_evaluations()} evaluations, ms per call {['%.2f' % t for t in times]}", flush=True)
d = np.abs(results["moments"] - results["passes"])
print(f"|dF| moments vs passes: max {d.max():.3e}, {int((d > 2e-6).sum())} of {G} genomes beyond 2e-6, {int((d == 0).sum())} identical")
bad = np.argsort(d)[-5:]
for g in bad:
    print(f"  genome {g}: start {start[g]:+.4f} moments {results['moments'][g]:+.8f} passes {results['passes'][g]:+.8f}")
m.close()
