"""Build small VCF-like record blocks (the input a parser would give PopulationDB) for parity tests.

Genotype codes come from the product's host generator (capi.synth_biallelic_host, bit-identical to the
HIP generator); loci decoration (offsets, bases) uses a seeded numpy generator.
"""
from __future__ import annotations

import numpy as np

from kgl_gene_amd import capi

from . import oracle_api as oa

BASES = np.array(list("ACGT"))


def genome_ids(n, prefix="HG"):
    return [f"{prefix}{i:06d}" for i in range(n)]


def biallelic_block(G, V, seed=1111, contig="Pf3D7_01_v3", max_gap=200, phased=True, rng_seed=7):
    """Returns (records, gt[V][G][2], codes[V][G], af[V]) for the synthetic biallelic population."""
    rows, af = capi.synth_biallelic_host(seed, 0, G, 0, V)
    codes = capi.unpack_dosage2(rows, G)
    rng = np.random.default_rng(rng_seed)
    offsets = np.cumsum(rng.integers(1, max_gap + 1, V)).astype(np.uint64)
    ref_i = rng.integers(0, 4, V)
    alt_i = (ref_i + rng.integers(1, 4, V)) % 4
    refs = BASES[ref_i].tolist()
    alts = [[a] for a in BASES[alt_i].tolist()]
    af6 = [np.tile(np.float32(a), (1, 6)) for a in af]
    rec = oa.Records(contig, offsets, refs, alts, af=af6)
    gt = np.zeros((V, G, 2), dtype=np.uint8)
    het_phase = rng.integers(0, 2, (V, G)).astype(np.uint8)
    gt[..., 0] = np.where(codes == 2, 1, np.where((codes == 1) & (het_phase == 0), 1, 0))
    gt[..., 1] = np.where(codes == 2, 1, np.where((codes == 1) & (het_phase == 1), 1, 0))
    return rec, gt, codes, af


def oracle_population(rec, gt, ids, mode):
    pop = oa.Population("synthetic")
    pop.add_genomes(ids, precreate=True)
    pop.add_records(rec, gt, mode)
    return pop


def variant_rows_in_reference_order(vdb: oa.VariantDB, rec: oa.Records):
    """Map the oracle's lexicographic-HGVS variant rank -> flat (record, alt) row index of the caller."""
    rec_idx, alt_idx = vdb.variant_keys()
    first_row = np.concatenate([[0], np.cumsum(rec.n_alts.astype(np.int64))[:-1]])
    return first_row[rec_idx.astype(np.int64)] + alt_idx.astype(np.int64)


def multiallelic_block(G, L, rng_seed=11, contig="chr1", indel_frac=0.15, missing_af_frac=0.02, dup_records=2):
    """Random mixed SNP/indel multiallelic loci (the shape of BASELINE config 4) as (records, gt[L][G][2]).

    ~70/20/10 % of loci have 1/2/3 alts; some alts are indels; a few AF values are missing; `dup_records`
    loci are repeated as a second record at the same offset (the only way a genome reaches > 2 copies)."""
    rng = np.random.default_rng(rng_seed)
    offsets, refs, alts, afs = [], [], [], []
    pos = 0
    bases = "ACGT"
    for _ in range(L):
        pos += int(rng.integers(1, 51))
        n_alt = int(rng.choice([1, 2, 3], p=[0.7, 0.2, 0.1]))
        ref = bases[rng.integers(0, 4)]
        cand = [b for b in bases if b != ref]
        rng.shuffle(cand)
        al = []
        for a in range(n_alt):
            if rng.random() < indel_frac:
                if rng.random() < 0.5:
                    al.append(ref + "".join(bases[i] for i in rng.integers(0, 4, int(rng.integers(1, 5)))))   # insertion
                else:
                    al.append("")   # placeholder: deletion needs a longer REF, patched below
            else:
                al.append(cand[a])
        if "" in al:
            ref_long = ref + "".join(bases[i] for i in rng.integers(0, 4, int(rng.integers(1, 5))))
            al = [ref if x == "" else (x + ref_long[1:] if len(x) == 1 else x + ref_long[1:]) for x in al]
            ref = ref_long
        # de-duplicate alts within the record
        seen, al2 = set(), []
        for x in al:
            while x in seen or x == ref:
                x = x + bases[rng.integers(0, 4)]
            seen.add(x)
            al2.append(x)
        p = rng.uniform(0.01, 0.5, n_alt)
        p *= min(1.0, 0.6 / p.sum())
        af = np.tile(p.astype(np.float32).reshape(-1, 1), (1, 6))
        af[rng.random(af.shape) < missing_af_frac] = np.nan
        offsets.append(pos); refs.append(ref); alts.append(al2); afs.append(af)
    for d in range(dup_records):
        i = int(rng.integers(0, L))
        offsets.append(offsets[i]); refs.append(refs[i]); alts.append(list(alts[i])); afs.append(afs[i].copy())
    R = len(offsets)
    rec = oa.Records(contig, np.array(offsets, dtype=np.uint64), refs, alts, af=afs)
    gt = np.zeros((R, G, 2), dtype=np.uint8)
    for r in range(R):
        p = np.nan_to_num(afs[r][:, 5].astype(np.float64), nan=0.05)
        probs = np.concatenate([[max(0.0, 1.0 - p.sum())], p])
        probs /= probs.sum()
        gt[r] = rng.choice(len(probs), size=(G, 2), p=probs).astype(np.uint8)
    return rec, gt


def synth_multiallelic_block(G, L, seed=1111, contig="chr1"):
    """The product's synthetic multi-allelic population (host twin of the device generator) as records + gt."""
    import ctypes as C

    gt8, table, alleles = capi.synth_multiallelic_host(seed, 0, G, 0, L)
    rng = np.random.default_rng(seed)
    offsets = np.cumsum(rng.integers(1, 51, L)).astype(np.uint64)
    refs, alts, afs = [], [], []
    bases = "ACGT"
    for l in range(L):
        n_alt = C.c_int(0)
        af = (C.c_float * 3)()
        indel = (C.c_int * 3)()
        capi.check(capi.lib().kgx_synth_locus_host(seed, l, C.byref(n_alt), af, indel))
        ref = bases[l % 4]
        cand = [b for b in bases if b != ref]
        al = []
        for a in range(n_alt.value):
            al.append(ref + "GA"[a % 2] * (a + 1) if indel[a] else cand[a])
        refs.append(ref)
        alts.append(al)
        afs.append(np.tile(np.array([af[a] for a in range(n_alt.value)], dtype=np.float32).reshape(-1, 1), (1, 6)))
    rec = oa.Records(contig, offsets, refs, alts, af=afs)
    return rec, alleles, gt8, table


def synth_multiallelic_coded(G, l0, l1, genome_base=0, seed=1111):
    """Loci [l0, l1) x genomes [genome_base, genome_base + G) of the product's synthetic multi-allelic population in the
    coded form of oracle_api.Population.add_records_coded (no Python strings: usable at millions of loci).
    Returns dict(offsets, ref_code, n_alts, alt_code, af_flat, alleles [n][G][2], gt8 [n][G], table [n][3])."""
    gt8, table, alleles = capi.synth_multiallelic_host(seed, genome_base, G, l0, l1)
    n_alt, af, indel = capi.synth_loci_host(seed, l0, l1)
    loci = np.arange(l0, l1, dtype=np.uint64)
    ref_code = (loci % 4).astype(np.uint8)
    cand = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=np.uint8)     # the bases that are not the reference's
    a_idx = np.arange(3, dtype=np.uint8)[None, :]
    codes = np.where(indel != 0, np.uint8(0x80) | a_idx, cand[ref_code])              # [n][3]
    keep = a_idx < n_alt[:, None]
    alt_code = codes[keep]
    af_flat = np.repeat(af[keep].astype(np.float32)[:, None], 6, axis=1)
    return dict(offsets=10 * loci + 1, ref_code=ref_code, n_alts=n_alt, alt_code=alt_code, af_flat=af_flat,
                alleles=alleles, gt8=gt8, table=table)
