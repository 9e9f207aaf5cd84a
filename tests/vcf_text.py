"""VCF text for the phased-diploid (1000 Genomes) flavour, built from a record block, with the token quirks the
reference's Genome1000VCFImpl::alternateIndex reacts to sprinkled in."""
from __future__ import annotations

import numpy as np


def write_vcf_1000(rec, gt, ids, rng_seed=0, quirks=True, contig=None):
    rng = np.random.default_rng(rng_seed)
    contig = contig or rec.contig
    lines = ["##fileformat=VCFv4.2", "##source=kgx-tests",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(ids)]
    for r in range(rec.n_records):
        alts = rec.alts[r]
        af = np.asarray(rec.af[r], dtype=np.float32).reshape(-1, 6)
        fields = ["AFR_AF", "AMR_AF", "EAS_AF", "EUR_AF", "SAS_AF", "AF"]
        info = []
        for sp, name in enumerate(fields):
            vals = ",".join("." if np.isnan(v) else repr(float(v)) for v in af[:, sp])
            info.append(f"{name}={vals}")
        q = int(rec.offsets[r])      # keyed by offset so that repeated records of one locus carry the same INFO
        if quirks and q % 37 == 5:
            info = [x for x in info if not x.startswith("AF=")]          # AF missing entirely
        if quirks and q % 41 == 7 and len(alts) > 1:
            info = [x if not x.startswith("AF=") else "AF=0.25" for x in info]   # scalar AF on a multi-alt record
        flt = "PASS" if r % 11 else ("pass" if r % 2 else "q10")
        cols = []
        for g in range(len(ids)):
            a, b = int(gt[r, g, 0]), int(gt[r, g, 1])
            tok = f"{a}|{b}"
            if quirks:
                u = rng.random()
                if u < 0.01:
                    tok = f".|{b}"
                elif u < 0.02:
                    tok = f"{a}|."
                elif u < 0.025:
                    tok = f"-|{b}"
                elif u < 0.03:
                    tok = f"{a}|-"
                elif u < 0.035:
                    tok = f"{a}/{b}"
                elif u < 0.04:
                    tok = f"{a}"
                elif u < 0.045:
                    tok = f"{len(alts) + 1}|{b}"
                elif u < 0.05:
                    tok = f"<CN2>|{b}"
                elif u < 0.055:
                    tok = f" {a}|{b} "
                elif u < 0.06:
                    tok = f"{a}|{b}|1"
                elif u < 0.065:
                    tok = ""
            cols.append(tok + (":35:12" if r % 3 == 0 else ""))
        ident = "." if r % 4 else f"rs{r}"
        lines.append("\t".join([contig, str(int(rec.offsets[r]) + 1), ident, rec.refs[r], ",".join(alts), ".", flt, ";".join(info),
                                "GT:GQ:DP" if r % 3 == 0 else "GT"] + cols))
    return "\n".join(lines) + "\n"
