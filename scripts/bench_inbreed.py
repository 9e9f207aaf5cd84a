"""C5-shaped inbreeding sweep timing on one GPU: python scripts/bench_inbreed.py [G] [L]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
m = capi.GenotypeMatrix(G, L)
t0 = time.perf_counter(); table = m.synth_multiallelic(1111, 0, 0); t_syn = time.perf_counter() - t0
alg_bytes = int(capi.lib().kgx_gt8_sweep_bytes(G, L, 3))
print(f"synth {t_syn:.2f}s; algorithmic bytes per frequency sweep: {alg_bytes/1e9:.2f} GB", flush=True)
for algo in ("Simple", "RitlandLocus", "HallME", "Loglikelihood"):
    iterative = algo in ("HallME", "Loglikelihood")
    if "--only-iterative" in sys.argv:
        if not iterative:
            continue
    elif iterative and G * L > 2e10 and "--all" not in sys.argv:
        continue
    t0 = time.perf_counter(); res = m.inbreed(table, algo, phased=True); dt_first = time.perf_counter() - t0
    t0 = time.perf_counter(); res = m.inbreed(table, algo, phased=True); dt = time.perf_counter() - t0
    ms = capi.inbreed_last_sweep_ms()
    print(f"{algo}: wall {dt*1e3:.1f} ms (first call {dt_first*1e3:.1f}; incl. H2D of the AF table, all passes); frequency sweep {ms:.2f} ms = {alg_bytes/ms/1e9:.2f} TB/s"
          f"  mean F {res['inbred_allele_sum'].mean():+.4f}" + (f"  evaluations {capi.inbreed_last_evaluations()}" if algo == "Loglikelihood" else ""), flush=True)
