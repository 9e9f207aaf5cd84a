"""The INBREED package's regime as the package issues it since round 4: WindowBatch windows x super populations per kgx_inbreed_batch
(one copy in, k_locus_tables + k_inbreed_window, one copy out) against the same tasks as kgx_inbreed calls.
usage: bench_inbreed_batch.py [windows per batch = 16]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
G, L, n_sel = 2560, 200_000, 1000
# five super populations of the 1000-Genomes sizes (661 AFR, 347 AMR, 504 EAS, 503 EUR, 489 SAS), each starting on a multiple of 16
sizes, ranges, at = (661, 347, 504, 503, 489), [], 0
for n in sizes:
    ranges.append((at, at + n))
    at = (at + n + 15) // 16 * 16
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
rng = np.random.default_rng(1)
tasks = []
for w in range(K):
    for g0, g1 in ranges:
        index = np.sort(rng.choice(L, n_sel - int(rng.integers(0, 40)), replace=False)).astype(np.uint32)      # every super population its own locus list
        tasks.append({"g0": g0, "g1": g1, "locus_index": index, "minor_af": np.ascontiguousarray(table[index])})
for _ in range(50):
    m.inbreed_batch(tasks, "Simple", phased=True)
for algo in ("Simple", "RitlandLocus", "HallME", "Loglikelihood"):
    for t in tasks:
        t["start"] = capi.reference_starts(algo, 4242, t["g1"] - t["g0"]) if algo in ("HallME", "Loglikelihood") else None
    m.inbreed_batch(tasks, algo, phased=True)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        batch = m.inbreed_batch(tasks, algo, phased=True)
    per_window = (time.perf_counter() - t0) / (reps * K)
    t0 = time.perf_counter()
    for t in tasks[:len(ranges) * 4]:
        single = m.inbreed(t["minor_af"], algo, phased=True, locus_index=t["locus_index"], g0=t["g0"], g1=t["g1"], start=t["start"])
    per_window_single = (time.perf_counter() - t0) / 4
    same = np.array_equal(batch[len(ranges) * 4 - 1]["total_allele_count"], single["total_allele_count"])
    print(f"{algo}: {per_window * 1e3:.3f} ms per window batched ({K} windows x {len(ranges)} super populations = {len(tasks)} tasks per batch), "
          f"{per_window_single * 1e3:.3f} ms per window as {len(ranges)} kgx_inbreed calls; counts equal: {same}", flush=True)
