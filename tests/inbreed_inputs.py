"""Test scaffolding: derive the inbreeding kernels' inputs (sampled loci, per-locus AF table, allele-index
bytes) from a VCF-like block, so the C ABI can be exercised directly.  The product's own version of this logic
is the C++ GpuInbreedAnalysis (kgl_gene_amd/csrc/host/), tested through the driver."""
from __future__ import annotations

import numpy as np


def is_snp(ref: str, alt: str) -> bool:
    if len(ref) == 1 and len(alt) == 1:
        return True
    if len(ref) != len(alt):
        return False
    return sum(1 for a, b in zip(ref, alt) if a != b) <= 1


class ReferenceLoci:
    """The SNP & PASS filtered reference contig: per offset the distinct (ref, alt) alts in record order."""

    def __init__(self, rec):
        self.offsets = []
        self.alts = []        # per locus: list of (ref, alt)
        self.af = []          # per locus: [n_alt][6] float32
        by_offset = {}
        for r in range(rec.n_records):
            if rec.passed is not None and not rec.passed[r]:
                continue
            for a, alt in enumerate(rec.alts[r]):
                if not is_snp(rec.refs[r], alt):
                    continue
                key = int(rec.offsets[r])
                lst = by_offset.setdefault(key, [])
                lst.append((rec.refs[r], alt, np.asarray(rec.af[r], dtype=np.float32).reshape(-1, 6)[a]))
        for off in sorted(by_offset):
            self.offsets.append(off)
            self.alts.append([(x[0], x[1]) for x in by_offset[off]])
            self.af.append(np.stack([x[2] for x in by_offset[off]]))
        self.offsets = np.array(self.offsets, dtype=np.uint64)

    def af_table(self, super_pop: int, amax: int):
        """[n_loci][amax] float64: AF of each alt for the super population; NaN where the alt is not in the
        AlleleFreqVector (no AF, or a duplicate of an earlier alt)."""
        t = np.full((len(self.offsets), amax), np.nan)
        for l, (alts, af) in enumerate(zip(self.alts, self.af)):
            seen = set()
            for j, key in enumerate(alts):
                f = af[j, super_pop]
                if np.isnan(f) or key in seen:
                    continue
                seen.add(key)
                t[l, j] = float(np.float32(f))
        return t

    def sample(self, table, lower, upper, spacing, min_af, max_af, count=None):
        """RetrieveLociiVector::getAllelesFromTo / getAllelesCount on the AF table: selected locus indices."""
        out = []
        prev = 0
        for l, off in enumerate(self.offsets):
            off = int(off)
            if off < lower:
                continue
            if count is None and off > upper:
                break
            if count is not None and len(out) >= count:
                break
            if not (off >= prev + spacing or prev == 0):
                continue
            f = np.clip(table[l][~np.isnan(table[l])], 0.0, 1.0)
            s = 0.0
            for x in f:
                s += x
            if len(f) == 0 or s - 1.0 > 1e-5:
                continue
            sc = min(max(s, 0.0), 1.0)
            if sc == 0.0 or sc < min_af or sc > max_af:
                continue
            prev = off
            out.append(l)
        return np.array(out, dtype=np.uint32)


def encode_gt8(rec, gt, loci: ReferenceLoci, phased_order=True):
    """[n_loci][G] bytes: the genome's SNP variants at each reference offset in OffsetDB order."""
    a1, a2 = encode_pairs(rec, gt, loci, phased_order, unknown=15, many=15)
    assert a1.max(initial=0) <= 15 and a2.max(initial=0) <= 15, "an offset with more than 14 alts: encode_wide"
    return (a1 | (a2 << 4)).astype(np.uint8)


def encode_wide(rec, gt, loci: ReferenceLoci, phased_order=True):
    """The same where offsets may hold more than 14 reference alts: (bytes [n_loci][G], wide locus indices, wide cells uint16
    [n_wide][G] = a1 | a2 << 8).  The byte row of a wide locus is 0xFF throughout (a call that reads bytes alone sees nothing there)."""
    a1, a2 = encode_pairs(rec, gt, loci, phased_order, unknown=255, many=255)
    wide = np.array([l for l, alts in enumerate(loci.alts) if len(alts) > 14], dtype=np.uint32)
    narrow1, narrow2 = np.where(a1 == 255, 15, a1), np.where(a2 == 255, 15, a2)
    assert np.delete(narrow1, wide, axis=0).max(initial=0) <= 15 and np.delete(narrow2, wide, axis=0).max(initial=0) <= 15
    out = (narrow1 | (narrow2 << 4)).astype(np.uint8)
    out[wide] = 0xFF
    cells = (a1[wide].astype(np.uint16) | (a2[wide].astype(np.uint16) << 8)).astype(np.uint16)
    return out, wide, cells


def encode_pairs(rec, gt, loci: ReferenceLoci, phased_order, unknown, many):
    """(a1, a2) [n_loci][G]: 1 + the place of the genome's first / second SNP variant in the offset's alt list, `unknown` for
    an alt the list does not hold, (many, many) for three or more variants."""
    R, G, _ = gt.shape
    index_of = {int(o): i for i, o in enumerate(loci.offsets)}
    carried = [[[] for _ in range(G)] for _ in range(len(loci.offsets))]
    order = [(0, None), (1, None)] if phased_order else None
    for r in range(R):
        off = int(rec.offsets[r])
        l = index_of.get(off)
        if l is None:
            continue
        keys = {}
        for j, key in enumerate(loci.alts[l]):
            keys.setdefault(key, j + 1)
        for g in range(G):
            for phase in range(2):
                a = int(gt[r, g, phase])
                if a == 0:
                    continue
                alt = rec.alts[r][a - 1]
                if not is_snp(rec.refs[r], alt):
                    continue
                code = keys.get((rec.refs[r], alt), unknown)
                carried[l][g].append((phase if phased_order else 0, r, phase, code))
    out1 = np.zeros((len(loci.offsets), G), dtype=np.uint32)
    out2 = np.zeros((len(loci.offsets), G), dtype=np.uint32)
    for l in range(len(loci.offsets)):
        for g in range(G):
            c = carried[l][g]
            if not c:
                continue
            if phased_order:
                # Genome1000 parser: per record all phase-A variants are added before phase-B ones
                c.sort(key=lambda t: (t[1], t[0]))
            else:
                c.sort(key=lambda t: (t[1], t[2]))
            if len(c) >= 3:
                out1[l, g] = out2[l, g] = many
            else:
                a1 = c[0][3]
                a2 = c[1][3] if len(c) == 2 else 0
                if phased_order and len(c) == 2 and a1 == a2 and a1 != unknown and c[0][2] == c[1][2]:
                    out1[l, g], out2[l, g] = 0, a1       # two copies on ONE phase (a repeated record): the (0, a) pair
                else:
                    out1[l, g], out2[l, g] = a1, a2
    return out1, out2
