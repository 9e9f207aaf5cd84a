"""VCF text for the phased-diploid (1000 Genomes) flavour, built from a record block, with the token quirks the
reference's Genome1000VCFImpl::alternateIndex reacts to sprinkled in; and for the unphased P. falciparum (Pf7)
flavour, with what PfVCFImpl::ParseRecord reacts to (GT separators, '*' alleles, spanning calls, AD shapes, tokens that
make std::stoll throw, alleles that need canonicalisation) and INFO fields on either side of P7VariantFilter's levels."""
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


def write_vcf_pf(n_records, ids, rng_seed=0, quirks=True, same_af_for_repeats=True, contigs=None):
    """Returns the text.  Several contigs (one of them the mitochondrion, one never used), multi-base alleles, repeated
    positions."""
    rng = np.random.default_rng(rng_seed)
    contigs = contigs or ["Pf3D7_01_v3", "Pf3D7_02_v3", "Pf3D7_MIT_v3", "Pf3D7_API_v3"]       # the last one is never used
    lines = ["##fileformat=VCFv4.2"] + [f"##contig=<ID={c},length=1000000>" for c in contigs] + [
        '##INFO=<ID=VQSLOD,Number=1,Type=Float,Description="x">',
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(ids)]
    bases = "ACGT"
    pos = 100
    for r in range(n_records):
        contig = contigs[int(rng.integers(0, len(contigs) - 1))]
        if r % 9 != 4:
            pos += int(rng.integers(1, 40))                      # r % 9 == 4 repeats the previous position
        shape = rng.random()
        ref = bases[int(rng.integers(0, 4))]
        if shape < 0.15:
            ref += "".join(bases[int(i)] for i in rng.integers(0, 4, int(rng.integers(1, 5))))
        n_alt = int(rng.integers(1, 4))
        alts = []
        for _ in range(n_alt):
            u = rng.random()
            if u < 0.55:
                alt = bases[(bases.index(ref[0]) + int(rng.integers(1, 4))) % 4] + ref[1:]      # SNP (or MNP-shaped SNP)
            elif u < 0.7:
                alt = ref + "".join(bases[int(i)] for i in rng.integers(0, 4, int(rng.integers(1, 4))))   # insert, shared prefix
            elif u < 0.8:
                alt = ref[0]                                                                       # delete when ref is long
            elif u < 0.9:
                alt = ref[:1] + "".join(bases[int(i)] for i in rng.integers(0, 4, int(rng.integers(1, 4)))) + ref[1:]
            elif u < 0.95 and quirks:
                alt = "*"
            else:
                alt = "".join(bases[int(i)] for i in rng.integers(0, 4, len(ref) + int(rng.integers(0, 3))))
            if alt == ref or alt in alts:
                alt = ref + "A" * (len(alts) + 1)
            alts.append(alt)
        info = []
        # keyed by the variant's canonical identity so that every record of one variant carries the same AF (the
        # reference bins each copy by its own record's AF; the product bins the row once)
        import zlib
        from . import oracle_api as oa
        salt = "" if same_af_for_repeats else f"#{r}"          # different AF per record: a repeated variant's copies land in different bins
        af = np.array([(zlib.crc32(f"{contig}:{oa.canonical(ref, a, pos - 1)}{salt}".encode()) % 6000) / 10000.0 for a in alts])
        if zlib.crc32(f"{contig}:{pos}".encode()) % 13 != 6:
            info.append("AF=" + ",".join(f"{x:.4f}" for x in af))
        u = rng.random()
        if u < 0.5:
            info.append(f"VQSLOD={rng.normal(1.0, 2.0):.3f}")
        elif u < 0.55:
            info.append("VQSLOD=.")
        elif u < 0.6:
            info.append("VQSLOD=nan")
        if rng.random() < 0.6:
            info.append(f"QD={rng.uniform(0.5, 20):.2f}")
        if rng.random() < 0.6:
            info.append(f"MQ={rng.uniform(3, 60):.2f}")
        if rng.random() < 0.6:
            info.append(f"SOR={rng.uniform(0.1, 4.5):.3f}")
        if rng.random() < 0.5:
            info.append(f"MQRankSum={rng.normal(-3, 6):.3f}")
        if rng.random() < 0.5:
            info.append(f"ReadPosRankSum={rng.normal(-2, 4):.3f}")
        info.append("DB")
        fmt = ["GT", "AD", "DP", "GQ"] if r % 5 else ["GT", "GQ", "AD"]
        if quirks and r % 53 == 17:
            fmt = ["GT", "DP"]                                    # no AD: the record is skipped
        cols = []
        for g in range(len(ids)):
            a, b = (int(x) for x in rng.choice(n_alt + 1, 2, p=[0.7] + [0.3 / n_alt] * n_alt))
            gt = f"{a}/{b}"
            ad = [int(x) for x in rng.integers(0, 30, n_alt + 1)]
            if rng.random() < 0.1:
                ad[0] = 0
                for k in range(1, n_alt + 1):
                    if rng.random() < 0.7:
                        ad[k] = 0                                 # spanning ("downstream") calls
            ad_text = ",".join(str(x) for x in ad)
            if quirks:
                u = rng.random()
                if u < 0.02:
                    gt = f"{a}|{b}"
                elif u < 0.04:
                    gt = "./."
                elif u < 0.05:
                    gt = "."
                elif u < 0.06:
                    gt = f"{a}"
                elif u < 0.065:
                    gt = f"{a}/{b}/1"
                elif u < 0.07:
                    gt = f"./{b}"                                 # first part not numeric: both stay reference
                elif u < 0.075 and r % 7 == 3:
                    gt = f"{a}/."                                 # std::stoll(".") throws: the rest of the record is lost
                elif u < 0.08:
                    ad_text = ",".join(str(x) for x in ad[:-1])   # wrong AD length
                elif u < 0.085:
                    ad_text = ad_text.replace(",", ",x", 1)       # a non-numeric depth is not stored
                elif u < 0.088 and r % 11 == 2:
                    ad_text = ad_text + ","                       # wrong length (empty trailing entry)
                elif u < 0.09:
                    gt = f"{n_alt + 1}/{b}"                       # index past the alt list
            vals = {"GT": gt, "AD": ad_text, "DP": str(sum(ad)), "GQ": str(int(rng.integers(0, 99)))}
            tok = ":".join(vals[k] for k in fmt)
            if quirks and rng.random() < 0.01:
                tok = gt                                          # truncated sample column: no AD field
            cols.append(tok)
        flt = "PASS" if r % 6 else "LowQual"
        lines.append("\t".join([contig, str(pos), ".", ref, ",".join(alts), "50", flt, ";".join(info), ":".join(fmt)] + cols))
    return "\n".join(lines) + "\n"


def write_vcf_mono(rec, source="Gnomad2_1", contig=None):
    """A mono-genome frequency source (Gnomad site file): no FORMAT / sample columns; the six super-population AF
    vectors under the INFO names the reference's table gives the data source."""
    from .records_io import AF_FIELDS

    fields = AF_FIELDS[source]
    contig = contig or rec.contig
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    for r in range(rec.n_records):
        af = np.asarray(rec.af[r], dtype=np.float32).reshape(-1, 6)
        info = []
        for sp, name in enumerate(fields):
            if name is None or any(x.startswith(name + "=") for x in info):
                continue
            info.append(f"{name}=" + ",".join("." if np.isnan(v) else repr(float(v)) for v in af[:, sp]))
        passed = True if rec.passed is None else bool(rec.passed[r])
        flt = ("PASS" if r % 3 else "pass") if passed else "AC0"
        lines.append("\t".join([contig, str(int(rec.offsets[r]) + 1), f"rs{r}", rec.refs[r], ",".join(rec.alts[r]), "100", flt, ";".join(info)]))
    return "\n".join(lines) + "\n"


def bgzip(data: bytes, block: int = 0xFF00, level: int = 6) -> bytes:
    """Block gzip as bgzip / htslib write it: gzip members of <= 64 KiB, each with the "BC" extra subfield holding its
    total size - 1, closed by the empty end-of-file block."""
    import struct
    import zlib

    def member(chunk: bytes) -> bytes:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        size = 12 + 6 + len(body) + 8
        head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, size - 1)
        return head + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))

    out = b"".join(member(data[i:i + block]) for i in range(0, len(data), block))
    return out + member(b"")
