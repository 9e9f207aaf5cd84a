"""HallME by moments against the 50 passes on a population the way sequenced cohorts look: ~95 % of the cells
reference-homozygous.  Biallelic loci, alt frequency q ~ U(0.005, 0.05); genotypes Hardy-Weinberg draws; 10 k genomes x
400 k loci (4 GB, loaded from the host)."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls (the library reads them at kgx_init / kgx_reload_options)

capi.init(0)
G, L = 10_000, 400_000
rng = np.random.default_rng(7)
q = rng.uniform(0.005, 0.05, L)
table = q.reshape(L, 1).copy()
m = capi.GenotypeMatrix(G, L)
block = 20_000
for l0 in range(0, L, block):                                  # bytes: 0x00 ref/ref, 0x01 alt on one phase, 0x11 alt/alt
    qq = q[l0:l0 + block, None]
    a = rng.random((len(qq), G), dtype=np.float32) < qq
    b = rng.random((len(qq), G), dtype=np.float32) < qq
    rows = (a.astype(np.uint8) | (b.astype(np.uint8) << 4))
    m.load_rows(rows, l0)
print(f"loaded; ref-hom fraction of the last block {float((rows == 0).mean()):.3f}", flush=True)
start = capi.reference_starts("HallME", 4242, G)
out = {}
for label, env in (("moments", None), ("50 passes", "1")):
    if env:
        os.environ["KGX_K7_HALL_PASSES"] = env
    m.inbreed(table, "HallME", phased=True, start=start)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = m.inbreed(table, "HallME", phased=True, start=start)
        walls.append(time.perf_counter() - t0)
    out[label] = res["inbred_allele_sum"].copy()
    print(f"HallME, {label}: {np.median(walls) * 1e3:.2f} ms per call ({G} x {L})  mean F {out[label].mean():+.6f}", flush=True)
print(f"|dF| moments vs 50 passes {np.abs(out['moments'] - out['50 passes']).max():.3e}")
