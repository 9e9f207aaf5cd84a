"""Ritland one-pass vs the generic kernel on the window-test population: which genomes / sums differ."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi
from tests import inbreed_inputs as ii, oracle_api as oa, synth_vcf as sv
capi.init(0)
G, L = 101, 1200
rec, gt = sv.multiallelic_block(G, L, rng_seed=5, missing_af_frac=0.03, dup_records=60)
loci = ii.ReferenceLoci(rec)
amax = max(len(a) for a in loci.alts)
bytes_ = ii.encode_gt8(rec, gt, loci, phased_order=True)
m = capi.GenotypeMatrix(G, len(loci.offsets))
m.load_rows(bytes_)
table = loci.af_table(oa.ALL, amax)
sel = loci.sample(table, 200, 40000, 60, 0.02, 0.9)
got = m.inbreed(table[sel], "RitlandLocus", phased=True, locus_index=sel)
os.environ["KGX_K5_GENERIC"] = "1"
ref = m.inbreed(table[sel], "RitlandLocus", phased=True, locus_index=sel)
print("amax", amax, "n_sel", len(sel))
for name in got.dtype.names:
    bad = np.flatnonzero(~np.isclose(got[name].astype(float), ref[name].astype(float), rtol=1e-12, atol=1e-12))
    print(name, "mismatch genomes:", bad[:10], (got[name][bad[:5]], ref[name][bad[:5]]) if len(bad) else "")
bad = np.flatnonzero(~np.isclose(got["major_hetero_freq"], ref["major_hetero_freq"], rtol=1e-12))
sub = bytes_[sel]
for g in bad[:3]:
    col = sub[:, g]
    vals, cnt = np.unique(col, return_counts=True)
    print("genome", g, "bytes:", {hex(v): int(c) for v, c in zip(vals, cnt)}, "diff", got["major_hetero_freq"][g] - ref["major_hetero_freq"][g])
    # which loci hold the unusual bytes and their table rows
    for s in np.flatnonzero((col > 0x33) | ((col & 0xF) == 0) & (col != 0))[:8]:
        print("   locus slot", s, "byte", hex(col[s]), "af row", table[sel][s])

# which odd cells were not taken off?  locus_class_frequencies columns: p_major, majorHom, majorHet, minorHom, minorHet
cf, valid = capi.locus_class_frequencies(table[sel], 0.0)
default = valid & (cf[:, 0] > 0.01)
print("valid", valid.sum(), "default", default.sum(), "of", len(sel))
t = table[sel]
for g in np.flatnonzero(~np.isclose(got["minor_hetero_freq"], ref["minor_hetero_freq"], rtol=1e-12))[:5]:
    col = sub[:, g]
    a1, a2 = (col & 15).astype(int), (col >> 4).astype(int)
    odd = []
    for s_ in np.flatnonzero(default & (col != 0)):
        outside = a1[s_] > amax or a2[s_] > amax
        nan1 = 0 < a1[s_] <= amax and np.isnan(t[s_, a1[s_] - 1])
        nan2 = 0 < a2[s_] <= amax and np.isnan(t[s_, a2[s_] - 1]) and not (a1[s_] == a2[s_])
        if outside or nan1 or (a1[s_] == 0 and 0 < a2[s_] <= amax and np.isnan(t[s_, a2[s_] - 1])) or (a2[s_] != 0 and a1[s_] != a2[s_] and nan2):
            odd.append(s_)
    odd = np.array(odd, dtype=int)
    want3 = cf[default, 4].sum() - cf[odd, 4].sum()
    print("genome", g, "got", got["minor_hetero_freq"][g], "ref", ref["minor_hetero_freq"][g], "numpy", want3, "all-default", cf[default, 4].sum())
    print("   odd loci", odd.tolist(), "bytes", [hex(col[s_]) for s_ in odd], "cf minorHet", np.round(cf[odd, 4], 4).tolist(), "cf majorHet", np.round(cf[odd, 2], 4).tolist())
