import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from kgl_gene_amd import capi
capi.WATCH_ENV = True
capi.init(0)
for G, L, amax in ((130, 20000, 14), (2049, 9000, 14), (5000, 9000, 3)):
    rng = np.random.default_rng(5)
    n_alts = rng.integers(1, amax + 1, L)
    weights = rng.gamma(0.35, 1.0, (L, amax)) * (np.arange(amax)[None, :] < n_alts[:, None])
    major = rng.uniform(0.45, 0.97, L)
    table = weights / weights.sum(axis=1, keepdims=True) * (1.0 - major)[:, None]
    table = np.where(np.arange(amax)[None, :] < n_alts[:, None], np.maximum(table, 2.0e-5), np.nan)
    cum = np.concatenate([major[:, None], major[:, None] + np.cumsum(np.nan_to_num(table), axis=1)], axis=1); cum /= cum[:, -1:]
    rows = np.zeros((L, G), dtype=np.uint8)
    for phase in range(2):
        allele = (rng.random((L, G))[:, :, None] >= cum[:, None, :]).sum(axis=2).astype(np.uint8)
        odd = rng.random((L, G)); allele = np.where(odd < 0.02, 15, allele).astype(np.uint8)
        rows |= allele << (4 * phase)
    m = capi.GenotypeMatrix(G, L); m.load_rows(rows)
    for phased in (True, False):
        start = capi.reference_starts("HallME", 77, G)
        got = m.inbreed(table, "HallME", phased=phased, start=start); path = capi.inbreed_last_path()
        os.environ["KGX_K7_HALL_PASSES"] = "1"
        want = m.inbreed(table, "HallME", phased=phased, start=start); os.environ.pop("KGX_K7_HALL_PASSES")
        d = np.abs(got["inbred_allele_sum"] - want["inbred_allele_sum"])
        same = all(np.array_equal(got[n], want[n]) for n in got.dtype.names if n != "inbred_allele_sum")
        print(G, L, amax, "phased" if phased else "unphased", path, capi.inbreed_last_path(), "counts equal", same, "max dF", float(d.max()), flush=True)
    m.close()
