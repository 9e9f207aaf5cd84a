"""VCF text -> columnar SoA, host code, CPU only: the product's direct flattener against the
oracle's restatement of the reference's phased-diploid parser (VCF text -> Variant objects -> PopulationDB ->
VariantDBVariant).  Same genomes, same variants in the same order, same dosage matrix, same INFO AF."""
import numpy as np
import pytest

from kgl_gene_amd import capi

from . import host_api as ha
from . import oracle_api as oa
from . import synth_vcf as sv
from . import vcf_text as vt


@pytest.mark.parametrize("contig", ["chr1", "chrX", "Y"])
def test_gt_token_table(contig):
    # the reference's decision table, token by token (oracle) == the product's parser driven through one-sample VCFs
    tokens = ["0|1", "1|0", "1|1", "2|1:34", "0|0", ".|1", "1|.", "-|1", "1|-", "1", "2", "1/1", "3|1", "<CN2>|1", "1|<CN0>",
              " 1|2 ", "", "0|1|2", "+1|1", "1x|2", "|1", "1|", "x|1", "1|x", "-1|1", "99999999999999999999999|1", ".", "-"]
    for tok in tokens:
        a, b = oa.gt_alternate_index(contig, tok, 2)
        text = f"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n{contig}\t100\t.\tA\tC,G\t.\tPASS\tAF=0.1,0.2\tGT\t{tok}\n"
        flat = ha.FlatVcf(text)
        want = {}
        for idx, alt in ((a, None), (b, None)):
            if idx:
                want[idx] = want.get(idx, 0) + 1
        got = {}
        for v, h in enumerate(flat.hgvs):
            alt_idx = 1 if h.endswith(">C") else 2
            got[alt_idx] = int(capi.unpack_dosage2(flat.packed[v:v + 1], flat.G)[0, 0])
        assert got == want, (tok, (a, b), got)


@pytest.mark.parametrize("threads", [1, 5])
def test_vcf_flatten_matches_oracle_parser(threads):
    G, L = 41, 900
    rec, gt = sv.multiallelic_block(G, L, rng_seed=77, dup_records=3)
    ids = [f"NA{i:05d}" for i in reversed(range(G))]                 # header order != sorted order
    text = vt.write_vcf_1000(rec, gt, ids, rng_seed=3)
    opop = oa.Population("vcf")
    n = opop.add_vcf_1000(text)
    assert n == rec.n_records
    vdb = oa.VariantDB(opop)
    flat = ha.FlatVcf(text, threads)
    assert flat.genome_ids == [vdb.genome_id(i) for i in range(vdb.n_genomes)]
    assert flat.hgvs == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    D = vdb.dosage()                                                 # [G][V] uint8 copies
    codes = capi.unpack_dosage2(flat.packed, flat.G)                 # [V][G]
    assert np.array_equal(codes, np.minimum(D.T, 3))
    assert flat.variant_objects == opop.variant_count() == int(D.sum())
    assert flat.non_diploid == int((D > 2).sum())
    assert flat.is_snp.sum() not in (0, flat.V)                      # both SNPs and indels present
    # FWS bins from the INFO AF column agree with the oracle's filter on its parsed population
    from kgl_gene_amd.fws import fws_bin_of_variant
    _, genome_out, _ = opop.fws()
    bins = fws_bin_of_variant(np.where(np.isinf(flat.info_af), np.nan, flat.info_af))
    dose = np.minimum(D.T, 3)
    for b in range(11):
        sel = dose[bins == b]
        want = np.stack([(sel == 0).sum(0), (sel == 1).sum(0), (sel == 2).sum(0)], 1)
        assert np.array_equal(genome_out[:, b, :], want.astype(np.uint64)), b


def test_vcf_flatten_edge_cases():
    hdr = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tB\tA\tC\n"
    # no records; blank lines; CRLF; a short line; a sample that carries nothing disappears (the parser never creates it)
    assert ha.FlatVcf(hdr).G == 0
    text = hdr + "\n" + "chr1\t10\t.\tA\tT\t.\tPASS\tAF=0.5\tGT\t0|1\t0|0\t1|1\r\n" + "chr1\t11\t.\n" + "chr1\t9\t.\tG\t.\t.\tPASS\t.\tGT\t0|0\t0|0\t0|0\n"
    flat = ha.FlatVcf(text)
    assert flat.genome_ids == ["B", "C"] and flat.hgvs == ["chr1:g.9A>T"]
    assert capi.unpack_dosage2(flat.packed, 2).tolist() == [[1, 2]]
    o = oa.Population("x")
    o.add_vcf_1000(text)
    vdb = oa.VariantDB(o)
    assert [vdb.hgvs(0)] == flat.hgvs and vdb.dosage().T.tolist() == [[1, 2]]


def test_canonical_sequences_through_one_record_vcfs():
    # Variant::canonicalSequences as restated in the oracle == what the product's Pf flattener writes into the HGVS
    cases = [("A", "T"), ("AT", "A"), ("A", "ATT"), ("ACG", "ATG"), ("ACGT", "ACCT"), ("ACGT", "AGT"), ("ACG", "ACGG"),
             ("AAAA", "AAA"), ("ACGTACGT", "ACGACGT"), ("TTTT", "TTTTTT"), ("ACG", "TGA"), ("AC", "AC"), ("ACGT", "AC"),
             ("CAG", "CAGCAG"), ("GATTACA", "GATACA"), ("AT", "TA"), ("ATG", "ACG"), ("AGG", "AG")]
    for ref, alt in cases:
        c_ref, c_alt, c_off = oa.canonical(ref, alt, 999)
        text = ("##contig=<ID=c1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
                f"c1\t1000\t.\t{ref}\t{alt}\t.\tPASS\t.\tGT:AD\t0/1:5,5\n")
        flat = ha.FlatVcf(text, flavour="Falciparum")
        assert flat.hgvs == [f"c1:g.{c_off}{c_ref}>{c_alt}"], (ref, alt, flat.hgvs)
        assert flat.offsets.tolist() == [c_off]
    # hand-checked: prefix keeps one base, suffix trimmed, offset moves with the prefix
    assert oa.canonical("ACGT", "ACCT", 10) == ("CG", "CC", 11)
    assert oa.canonical("ACGTACGT", "ACGACGT", 0) == ("GT", "G", 2)
    assert oa.canonical("AT", "A", 5) == ("AT", "A", 5)


@pytest.mark.parametrize("threads,quality_filter", [(1, False), (5, False), (3, True)])
def test_vcf_pf_flatten_matches_oracle_parser(threads, quality_filter):
    G, L = 29, 1500
    ids = [f"PF{i:04d}-C" for i in reversed(range(G))]
    text = vt.write_vcf_pf(L, ids, rng_seed=11)
    opop = oa.Population("pf")
    assert opop.add_vcf_pf(text) == L
    if quality_filter:
        full_count = opop.variant_count()
        opop = opop.filter_p7()
        assert 0 < opop.variant_count() < full_count              # the filter bites, and not everything
    vdb = oa.VariantDB(opop)
    flat = ha.FlatVcf(text, threads, flavour="Falciparum", quality_filter=quality_filter)
    assert flat.genome_ids == sorted(ids) == [vdb.genome_id(i) for i in range(vdb.n_genomes)]   # every sample is a genome
    assert flat.hgvs == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    D = vdb.dosage()
    codes = capi.unpack_dosage2(flat.packed, flat.G)
    assert np.array_equal(codes, np.minimum(D.T, 3))
    assert flat.variant_objects == opop.variant_count() == int(D.sum())
    assert flat.non_diploid == int((D > 2).sum())
    assert 0 < flat.is_snp.sum() < flat.V
    assert not any(h.endswith(">*") for h in flat.hgvs)             # the upstream-deletion allele never becomes a Variant
    from kgl_gene_amd.fws import fws_bin_of_variant
    _, genome_out, _ = opop.fws()
    bins = fws_bin_of_variant(np.where(np.isinf(flat.info_af), np.nan, flat.info_af))
    dose = np.minimum(D.T, 3)
    for b in range(11):
        sel = dose[bins == b]
        want = np.stack([(sel == 0).sum(0), (sel == 1).sum(0), (sel == 2).sum(0)], 1)
        assert np.array_equal(genome_out[:, b, :], want.astype(np.uint64)), b


def test_vcf_pf_edge_cases():
    hdr = "##contig=<ID=c1,length=5>\n##contig=<ID=c2>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tB\tA\n"
    flat = ha.FlatVcf(hdr, flavour="Falciparum")
    assert flat.genome_ids == ["A", "B"] and flat.V == 0                 # genomes exist without any record
    body = ("c1\t10\t.\tA\tT,*\t.\tPASS\tVQSLOD=-1\tGT:AD\t1/2:3,4,5\t0/1:0,0,9\n"     # B: T once ('*' dropped); A: spanning call
            "c1\t11\t.\tG\tC\t.\tPASS\tQD=1.0\tGT:AD\t1/1:1,8\t1/.:2,2\n"              # A: stoll('.') throws after B was added
            "c1\t12\t.\tG\tC\t.\tPASS\t.\tGT\t1/1\t1/1\n")                            # no AD in FORMAT
    flat = ha.FlatVcf(hdr + body, flavour="Falciparum")
    assert flat.hgvs == ["c1:g.10G>C", "c1:g.9A>T"]
    assert capi.unpack_dosage2(flat.packed, 2).tolist() == [[0, 2], [0, 1]]
    o = oa.Population("x")
    o.add_vcf_pf(hdr + body)
    vdb = oa.VariantDB(o)
    assert [vdb.hgvs(i) for i in range(vdb.n_variants)] == flat.hgvs and vdb.dosage().T.tolist() == [[0, 2], [0, 1]]
    # both records fail P7VariantFilter (VQSLOD < 0; QD < 2)
    assert ha.FlatVcf(hdr + body, flavour="Falciparum", quality_filter=True).V == 0
    assert o.filter_p7().variant_count() == 0


def test_missing_alt_is_the_empty_allele_in_every_flavour():
    """ParseVCF::moveToVcfRecord (kgl_variant_vcf_impl.cpp:133-141) turns an ALT of "." into "" before any parser sees
    it: the variant is REF>"" -- a deletion, not a SNP -- in the mono-genome, Pf and 1000-Genomes flavours alike."""
    from .records_io import DATA_SOURCE

    mono = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n1\t10\t.\tA\t.\t.\tPASS\tAF=0.25\n1\t20\t.\tC\tT\t.\tPASS\tAF=0.5\n"
    dip = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n1\t10\t.\tA\t.\t.\tPASS\t.\tGT\t1|0\n1\t20\t.\tC\tT\t.\tPASS\t.\tGT\t1|1\n"
    o = oa.Population("mono")
    o.add_vcf_mono(mono, "Gnomad2_1")
    vdb = oa.VariantDB(o)
    assert sorted(vdb.hgvs(i) for i in range(vdb.n_variants)) == ["1:g.19C>T", "1:g.9A>"]
    assert o.filter_snp_pass().variant_count() == 1                      # "A" > "" is not a SNP
    got = ha.InbreedInputs(mono, DATA_SOURCE["Gnomad2_1"], dip)
    assert got.error == "" and got.offsets.tolist() == [19]
    flat = ha.FlatVcf(dip)
    assert flat.hgvs == ["1:g.19C>T", "1:g.9A>"]
    pf = "##contig=<ID=1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n1\t10\t.\tAC\t.\t.\tPASS\t.\tGT:AD\t0/1:3,4\n"
    o = oa.Population("pf")
    o.add_vcf_pf(pf)
    vdb = oa.VariantDB(o)
    flat = ha.FlatVcf(pf, flavour="Falciparum")
    assert [vdb.hgvs(i) for i in range(vdb.n_variants)] == flat.hgvs and len(flat.hgvs) == 1


@pytest.mark.parametrize("threads", [1, 4])
def test_inbreed_inputs_from_vcf_match_the_scaffold_encoder(threads):
    """Reference site VCF + 1000-Genomes VCF -> (reference loci, AF per super population, allele-index bytes): the
    product's flatteners against tests/inbreed_inputs.py, the independent numpy restatement the kernel tests use."""
    from . import inbreed_inputs as ii
    from .records_io import DATA_SOURCE

    G, L = 31, 1200
    rec, gt = sv.multiallelic_block(G, L, rng_seed=5, missing_af_frac=0.05, dup_records=0)
    for a in rec.af:
        a[:, 4] = a[:, 5]
    ids = [f"NA{i:05d}" for i in reversed(range(G))]
    ref_text = vt.write_vcf_mono(rec, "Gnomad2_1")
    dip_text = vt.write_vcf_1000(rec, gt, ids, rng_seed=1, quirks=False)
    got = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], dip_text, threads)
    assert got.error == "" and got.contigs == 1
    loci = ii.ReferenceLoci(rec)
    assert np.array_equal(got.offsets, loci.offsets)
    assert got.n_alts.tolist() == [len(a) for a in loci.alts] and got.amax == max(len(a) for a in loci.alts)
    for l in range(got.L):
        want = np.asarray(loci.af[l], dtype=np.float64)
        assert np.array_equal(np.isnan(got.af[l, :len(want)]), np.isnan(want)) and np.allclose(np.nan_to_num(got.af[l, :len(want)]), np.nan_to_num(want), rtol=0, atol=0)
    # genomes: carriers of anything on the contig, in id order
    carriers = sorted(ids[g] for g in range(G) if gt[:, g, :].any())
    assert got.genome_ids == carriers
    column = {name: g for g, name in enumerate(ids)}
    want_bytes = ii.encode_gt8(rec, gt, loci)[:, [column[name] for name in carriers]]
    assert np.array_equal(got.bytes, want_bytes)
    # the population from a file read in pieces (a line or so, a few lines, everything): the same bytes
    if threads == 1:
        import tempfile
        from pathlib import Path

        with tempfile.TemporaryDirectory() as tmp:
            for name, payload in (("plain.vcf", dip_text.encode()), ("block.vcf.bgz", vt.bgzip(dip_text.encode(), block=3000))):
                (Path(tmp) / name).write_bytes(payload)
                for chunk_bytes in (1, 2500, 0):
                    piecewise = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / name, chunk_bytes=chunk_bytes)
                    assert piecewise.genome_ids == got.genome_ids and np.array_equal(piecewise.bytes, got.bytes), (name, chunk_bytes)
                    # ... and streamed: rows handed to a sink as their loci are complete, one locus' records held between pieces
                    if len(carriers) == G:
                        streamed = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / name, chunk_bytes=chunk_bytes,
                                                    streaming=True)
                        assert streamed.genome_ids == got.genome_ids and np.array_equal(streamed.bytes, got.bytes), (name, chunk_bytes)
            assert len(carriers) == G
            # repeated records right behind their first (a sorted file): the streamed rows are the two-phase ones; at the end of the
            # file (positions no longer ascending), a sample without any variant, a sample named twice: the two-phase path is asked for
            lines = dip_text.split("\n")
            header = next(i for i, ln in enumerate(lines) if ln.startswith("#CHROM"))
            body = [ln for ln in lines[header + 1:] if ln]
            doubled = []
            for k, ln in enumerate(body):
                doubled.append(ln)
                if k % 9 == 2:
                    doubled.append(ln)
            adjacent = "\n".join(lines[:header + 1] + doubled) + "\n"
            (Path(tmp) / "adjacent.vcf").write_text(adjacent)
            want_adjacent = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], adjacent, 2)
            for chunk_bytes in (1, 3000, 0):
                streamed = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / "adjacent.vcf", chunk_bytes=chunk_bytes, streaming=True)
                assert np.array_equal(streamed.bytes, want_adjacent.bytes) and not np.array_equal(streamed.bytes, got.bytes), chunk_bytes
            (Path(tmp) / "late.vcf").write_text("\n".join(lines[:header + 1] + body + body[5:8]) + "\n")
            with pytest.raises(ha.TwoPhaseNeeded, match="ascending"):
                ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / "late.vcf", chunk_bytes=2000, streaming=True)
            silent = ["\t".join(ln.split("\t")[:9] + ["0|0"] + ln.split("\t")[10:]) for ln in body]
            (Path(tmp) / "silent.vcf").write_text("\n".join(lines[:header + 1] + silent) + "\n")
            with pytest.raises(ha.TwoPhaseNeeded, match="carries no variant"):
                ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / "silent.vcf", streaming=True)
            cols = lines[header].split("\t")
            cols[-1] = cols[-2]
            (Path(tmp) / "twice.vcf").write_text("\n".join(lines[:header] + ["\t".join(cols)] + body) + "\n")
            with pytest.raises(ha.TwoPhaseNeeded, match="named twice"):
                ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / "twice.vcf", streaming=True)
            # and the reference site file in pieces as well
            (Path(tmp) / "sites.vcf.bgz").write_bytes(vt.bgzip(ref_text.encode(), block=5000))
            for chunk_bytes in (1, 4000, 0):
                both = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=Path(tmp) / "plain.vcf", chunk_bytes=chunk_bytes,
                                        reference_path=Path(tmp) / "sites.vcf.bgz")
                assert np.array_equal(both.offsets, got.offsets) and np.array_equal(both.af, got.af, equal_nan=True) and np.array_equal(both.bytes, got.bytes)
            with pytest.raises(IOError):
                ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, diploid_path=Path(tmp) / "missing.vcf")
    # a repeated record gives its carriers a second copy on the SAME phase: the (0, a) byte, as the scaffold encoder writes it
    l0 = next(l for l in range(got.L) if (got.bytes[l] & 0xF).any() and ((got.bytes[l] & 0xF) != 15).any())
    pos = str(int(got.offsets[l0]) + 1)
    repeated = [line for line in dip_text.split("\n") if line and not line.startswith("#") and line.split("\t")[1] == pos]
    again = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], dip_text + "\n".join(repeated) + "\n")
    assert again.error == ""
    changed = again.bytes[l0] != got.bytes[l0]
    assert changed.any() and np.array_equal(again.bytes[np.arange(got.L) != l0], got.bytes[np.arange(got.L) != l0])
    before, after = got.bytes[l0][changed], again.bytes[l0][changed]
    single = (before >> 4) == 0                                      # one copy before -> the same allele twice on its phase
    assert np.array_equal(after[single], (before[single] & 0xF) << 4)
    assert np.all(after[~single] == 0xFF)                            # two copies before -> four now: ">= 3 variants"


@pytest.mark.parametrize("threads", [1, 6])
def test_vcf_reader_plain_gzip_and_block_gzip(tmp_path, threads):
    import gzip

    ids = [f"PF{i:04d}-C" for i in range(7)]
    text = vt.write_vcf_pf(4000, ids, rng_seed=2).encode()        # a few hundred KiB: several bgzf blocks
    assert len(text) > 3 * 0xFF00
    (tmp_path / "a.vcf").write_bytes(text)
    (tmp_path / "a.vcf.gz").write_bytes(gzip.compress(text[:100000]) + gzip.compress(text[100000:]))   # two members
    (tmp_path / "a.vcf.bgz").write_bytes(vt.bgzip(text))
    (tmp_path / "tiny.bgz").write_bytes(vt.bgzip(b""))
    for name in ("a.vcf", "a.vcf.gz", "a.vcf.bgz"):
        assert ha.read_vcf_text(tmp_path / name, threads) == text, name
    assert ha.read_vcf_text(tmp_path / "tiny.bgz", threads) == b""
    # a flipped byte inside a block's data fails that block's CRC / size check; a truncated gzip is refused
    bad = bytearray(vt.bgzip(text))
    bad[len(bad) // 2] ^= 0x55
    (tmp_path / "bad.bgz").write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        ha.read_vcf_text(tmp_path / "bad.bgz", threads)
    (tmp_path / "cut.gz").write_bytes(gzip.compress(text)[:-20])
    with pytest.raises(ValueError):
        ha.read_vcf_text(tmp_path / "cut.gz", threads)
    with pytest.raises(ValueError):
        ha.read_vcf_text(tmp_path / "missing.vcf", threads)
    # and the flattener sees the same population through any of them
    flat = ha.FlatVcf(ha.read_vcf_text(tmp_path / "a.vcf.bgz", threads).decode(), flavour="Falciparum")
    assert flat.G == len(ids) and flat.V > 0


def same_flat(a, b):
    return (a.hgvs == b.hgvs and a.genome_ids == b.genome_ids and np.array_equal(a.packed, b.packed) and a.variant_objects == b.variant_objects
            and a.non_diploid == b.non_diploid and np.array_equal(a.info_af, b.info_af, equal_nan=True) and np.array_equal(a.split_packed, b.split_packed)
            and np.array_equal(a.split_of, b.split_of) and np.array_equal(a.from_splits, b.from_splits))


@pytest.mark.parametrize("flavour", ["Genome1000", "Falciparum"])
def test_streaming_flatten_equals_the_two_phase_one(tmp_path, flavour):
    """The streaming flatteners hand rows to a sink piece by piece while the file is read (first-appearance order, repeated
    records merged into their row at the end, split rows last); put back into HGVS order the population is the two-phase
    flatteners' -- rows, split rows, non-diploid cells, Variant-object count -- whatever the piece size.  Files they cannot
    take say so instead of guessing."""
    if flavour == "Genome1000":
        G, L = 37, 900
        rec, gt = sv.multiallelic_block(G, L, rng_seed=9, dup_records=40)
        for i in range(L, L + 40, 2):
            rec.af[i] = rec.af[i] * np.float32(0.3)                        # half of the repeats in another FWS bin than their first record
        by_offset = np.argsort(rec.offsets, kind="stable")                 # repeated records right behind their first, as in a sorted VCF
        sorted_rec = oa.Records(rec.contig, rec.offsets[by_offset], [rec.refs[i] for i in by_offset], [rec.alts[i] for i in by_offset],
                                af=[rec.af[i] for i in by_offset])
        ids = [f"HG{i:05d}" for i in reversed(range(G))]
        text = vt.write_vcf_1000(sorted_rec, gt[by_offset], ids, rng_seed=2)
        unsorted_text = vt.write_vcf_1000(rec, gt, ids, rng_seed=2)
    else:
        text = vt.write_vcf_pf(700, [f"PF{i:04d}-C" for i in range(19)], rng_seed=5, same_af_for_repeats=False)
    whole = ha.FlatVcf(text, 3, flavour=flavour, quality_filter=(flavour == "Falciparum"))
    assert whole.V > 300 and whole.n_split > 0 and (whole.non_diploid > 0 or flavour == "Falciparum")
    data = text.encode()
    for kind, payload in {"plain": data, "bgzf": vt.bgzip(data, block=4000)}.items():
        path = tmp_path / f"population.{kind}"
        path.write_bytes(payload)
        for chunk_bytes in (1, 777, 9000, 0):
            got = ha.FlatVcf(None, 2, flavour=flavour, quality_filter=(flavour == "Falciparum"), path=path, chunk_bytes=chunk_bytes, streaming=True)
            assert same_flat(got, whole), (kind, chunk_bytes)
    if flavour == "Genome1000":
        (tmp_path / "unsorted.vcf").write_text(unsorted_text)              # the repeats at the end of the file, pieces after their first record
        assert same_flat(ha.FlatVcf(None, 2, path=tmp_path / "unsorted.vcf", chunk_bytes=5000, streaming=True), ha.FlatVcf(unsorted_text, 2))
        lines = text.split("\n")
        header = next(i for i, ln in enumerate(lines) if ln.startswith("#CHROM"))
        silent = [ln if i <= header or not ln else "\t".join(ln.split("\t")[:9] + ["0|0"] + ln.split("\t")[10:]) for i, ln in enumerate(lines)]
        (tmp_path / "silent.vcf").write_text("\n".join(silent))            # the first sample carries nothing: it is no genome
        with pytest.raises(ha.TwoPhaseNeeded, match="carries no variant"):
            ha.FlatVcf(None, 2, path=tmp_path / "silent.vcf", streaming=True)
        assert ha.FlatVcf(None, 2, path=tmp_path / "silent.vcf").G == G - 1
        twice = list(lines)
        cols = twice[header].split("\t")
        cols[-1] = cols[-2]
        twice[header] = "\t".join(cols)
        (tmp_path / "twice.vcf").write_text("\n".join(twice))
        with pytest.raises(ha.TwoPhaseNeeded, match="named twice"):
            ha.FlatVcf(None, 2, path=tmp_path / "twice.vcf", streaming=True)
    (tmp_path / "empty").write_bytes(b"")
    assert ha.FlatVcf(None, 1, flavour=flavour, path=tmp_path / "empty", streaming=True).V == 0


@pytest.mark.parametrize("flavour", ["Genome1000", "Falciparum"])
def test_flatten_from_file_in_pieces_equals_flatten_of_the_text(tmp_path, flavour):
    """The file entry points read a bounded piece of whole lines at a time (plain, gzip, block gzip); whatever the piece
    size -- smaller than a line, a few lines, the whole file -- the population is the one the whole text flattens to."""
    import gzip

    if flavour == "Genome1000":
        G, L = 37, 900
        rec, gt = sv.multiallelic_block(G, L, rng_seed=9, dup_records=12)
        text = vt.write_vcf_1000(rec, gt, [f"HG{i:05d}" for i in reversed(range(G))], rng_seed=2)
    else:
        text = vt.write_vcf_pf(500, [f"PF{i:04d}-C" for i in range(19)], rng_seed=5)
    whole = ha.FlatVcf(text, 3, flavour=flavour, quality_filter=(flavour == "Falciparum"))
    assert whole.V > 300
    data = text.encode()
    files = {"plain": data, "gzip": gzip.compress(data[:len(data) // 3]) + gzip.compress(data[len(data) // 3:]), "bgzf": vt.bgzip(data, block=4000)}
    for kind, payload in files.items():
        path = tmp_path / f"population.{kind}"
        path.write_bytes(payload)
        for chunk_bytes in (1, 777, 50_000, 0):
            got = ha.FlatVcf(None, 2, flavour=flavour, quality_filter=(flavour == "Falciparum"), path=path, chunk_bytes=chunk_bytes)
            assert same_flat(got, whole), (kind, chunk_bytes)
    rng = np.random.default_rng(12)
    for kind in files:                                  # and random piece sizes, from a few bytes to several lines
        for chunk_bytes in rng.integers(2, 30_000, 6):
            got = ha.FlatVcf(None, 3, flavour=flavour, quality_filter=(flavour == "Falciparum"), path=tmp_path / f"population.{kind}", chunk_bytes=int(chunk_bytes))
            assert same_flat(got, whole), (kind, int(chunk_bytes))
    # no final newline; an empty file; a truncated block-gzip file; a missing file
    (tmp_path / "open_end").write_bytes(data.rstrip(b"\n"))
    assert same_flat(ha.FlatVcf(None, 2, flavour=flavour, quality_filter=(flavour == "Falciparum"), path=tmp_path / "open_end", chunk_bytes=3000), whole)
    (tmp_path / "empty").write_bytes(b"")
    assert ha.FlatVcf(None, 1, flavour=flavour, path=tmp_path / "empty").V == 0
    (tmp_path / "cut.bgz").write_bytes(files["bgzf"][:len(files["bgzf"]) // 2])
    with pytest.raises(IOError):
        ha.FlatVcf(None, 1, flavour=flavour, path=tmp_path / "cut.bgz", chunk_bytes=5000)
    (tmp_path / "cut.gz").write_bytes(files["gzip"][:len(files["gzip"]) // 2])
    with pytest.raises(IOError):
        ha.FlatVcf(None, 1, flavour=flavour, path=tmp_path / "cut.gz", chunk_bytes=5000)
    with pytest.raises(IOError):
        ha.FlatVcf(None, 1, flavour=flavour, path=tmp_path / "not_there")


def test_repeated_records_in_different_fws_bins_are_split_per_bin():
    """CalcFWS filters Variant objects by their own record's AF: when the records of one variant disagree, each bin's
    population holds only that record's copies.  The flatteners emit per-bin split rows for exactly those variants."""
    hdr = "##contig=<ID=c1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\tS3\n"
    body = ("c1\t10\t.\tA\tT\t.\tPASS\tAF=0.02\tGT:AD\t0/1:5,5\t0/0:5,0\t1/1:0,9\n"      # bin 0
            "c1\t10\t.\tA\tT\t.\tPASS\tAF=0.22\tGT:AD\t0/1:5,5\t0/1:5,5\t0/0:9,0\n"      # same variant, bin 4
            "c1\t10\t.\tA\tT\t.\tPASS\t.\tGT:AD\t0/0:5,0\t0/1:5,5\t0/0:9,0\n"            # same variant, no AF: in no bin
            "c1\t20\t.\tG\tC\t.\tPASS\tAF=0.31\tGT:AD\t0/1:5,5\t0/0:5,0\t0/0:9,0\n"
            "c1\t20\t.\tG\tC\t.\tPASS\tAF=0.33\tGT:AD\t0/0:5,0\t1/1:0,5\t0/0:9,0\n")     # same variant, SAME bin 6: no split
    flat = ha.FlatVcf(hdr + body, flavour="Falciparum")
    assert flat.hgvs == ["c1:g.19G>C", "c1:g.9A>T"] and flat.from_splits.tolist() == [0, 1] and flat.n_split == 2
    assert capi.unpack_dosage2(flat.packed, 3).tolist() == [[1, 2, 0], [2, 2, 2]]                 # merged rows: every copy
    assert sorted(zip([round(float(x), 2) for x in flat.split_info_af], capi.unpack_dosage2(flat.split_packed, 3).tolist())) == \
        [(0.02, [1, 0, 2]), (0.22, [1, 1, 0])]
    o = oa.Population("x")
    o.add_vcf_pf(hdr + body)
    _, genome_out, _ = o.fws()
    assert np.array_equal(ha.fws_genome_bins(flat), genome_out)
    assert genome_out[:, 0, :].tolist() == [[0, 1, 0], [1, 0, 0], [0, 0, 1]]                      # bin 0 sees the first record only


@pytest.mark.parametrize("flavour", ["Falciparum", "Genome1000"])
def test_fws_bins_with_disagreeing_repeats_match_oracle(flavour):
    if flavour == "Falciparum":
        ids = [f"PF{i:04d}-C" for i in range(19)]
        text = vt.write_vcf_pf(2500, ids, rng_seed=31, same_af_for_repeats=False)
        o = oa.Population("pf")
        o.add_vcf_pf(text)
    else:
        G, L = 23, 900
        rec, gt = sv.multiallelic_block(G, L, rng_seed=12, dup_records=40)
        rng = np.random.default_rng(5)
        for a in rec.af:                                      # every record its own AF: repeats of a locus disagree
            a[:, 5] = rng.uniform(0, 0.6, a.shape[0]).astype(np.float32)
        text = vt.write_vcf_1000(rec, gt, [f"NA{i:05d}" for i in range(G)], rng_seed=3, quirks=False)
        o = oa.Population("kg")
        o.add_vcf_1000(text)
    flat = ha.FlatVcf(text, 3, flavour=flavour)
    assert flat.n_split > 0 and flat.from_splits.sum() > 0
    _, genome_out, vdb = o.fws()
    assert flat.hgvs == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert np.array_equal(ha.fws_genome_bins(flat), genome_out)


@pytest.mark.parametrize("seed", range(6))
def test_flatteners_vs_oracle_over_seeds(seed):
    """More draws of both VCF flavours (token quirks on, repeated records with their own AF, duplicate sample names)
    through the product's flatteners and the oracle's parsers: same genomes, variants, dosages, FWS bins."""
    rng = np.random.default_rng(1000 + seed)
    # 1000-Genomes flavour
    G, L = int(rng.integers(3, 40)), int(rng.integers(50, 600))
    rec, gt = sv.multiallelic_block(G, L, rng_seed=200 + seed, dup_records=int(rng.integers(0, 30)))
    for a in rec.af:
        a[:, 5] = rng.uniform(0, 0.6, a.shape[0]).astype(np.float32)
    ids = [f"NA{i:05d}" for i in rng.permutation(G)]
    if G > 4:
        ids[1] = ids[0]                                         # one genome named twice: its columns add up
    text = vt.write_vcf_1000(rec, gt, ids, rng_seed=seed, quirks=True)
    o = oa.Population("kg")
    o.add_vcf_1000(text)
    flat = ha.FlatVcf(text, 1 + seed % 4)
    _, genome_out, vdb = o.fws()
    assert flat.genome_ids == [vdb.genome_id(i) for i in range(vdb.n_genomes)]
    assert flat.hgvs == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert np.array_equal(capi.unpack_dosage2(flat.packed, flat.G), np.minimum(vdb.dosage().T, 3))
    assert np.array_equal(ha.fws_genome_bins(flat), genome_out)
    # P. falciparum flavour, with and without the record filter
    ids = [f"PF{i:04d}-C" for i in rng.permutation(int(rng.integers(2, 30)))]
    text = vt.write_vcf_pf(int(rng.integers(100, 1200)), ids, rng_seed=300 + seed, same_af_for_repeats=bool(seed % 2))
    for quality_filter in (False, True):
        o = oa.Population("pf")
        o.add_vcf_pf(text)
        if quality_filter:
            o = o.filter_p7()
        flat = ha.FlatVcf(text, 1 + seed % 3, flavour="Falciparum", quality_filter=quality_filter)
        _, genome_out, vdb = o.fws()
        assert flat.genome_ids == [vdb.genome_id(i) for i in range(vdb.n_genomes)]
        assert flat.hgvs == [vdb.hgvs(i) for i in range(vdb.n_variants)]
        assert np.array_equal(capi.unpack_dosage2(flat.packed, flat.G), np.minimum(vdb.dosage().T, 3))
        assert np.array_equal(ha.fws_genome_bins(flat), genome_out)


def test_flatteners_survive_mangled_text():
    """Robustness, not parity: truncated lines, deleted tabs, random bytes, missing header lines.  The flatteners must
    return (possibly empty) populations, never crash or hang (tests/tools/sanitize_host.sh runs this under ASan/UBSan)."""
    from .records_io import DATA_SOURCE

    rng = np.random.default_rng(99)
    rec, gt = sv.multiallelic_block(9, 120, rng_seed=4, dup_records=5)
    ids = [f"NA{i:05d}" for i in range(9)]
    texts = [vt.write_vcf_1000(rec, gt, ids, rng_seed=1), vt.write_vcf_pf(150, [f"PF{i}" for i in range(7)], rng_seed=6),
             vt.write_vcf_mono(rec, "Gnomad2_1")]
    for trial in range(60):
        base = texts[trial % 3]
        raw = bytearray(base.encode())
        kind = trial % 5
        if kind == 0:                                            # random byte flips (printable range)
            for pos in rng.integers(0, len(raw), 40):
                raw[pos] = int(rng.integers(9, 127))
        elif kind == 1:                                          # drop tabs
            for pos in [i for i, c in enumerate(raw) if c == 9][:: int(rng.integers(3, 40))]:
                raw[pos] = ord(" ")
        elif kind == 2:                                          # truncate in the middle of a line
            raw = raw[: int(rng.integers(1, len(raw)))]
        elif kind == 3:                                          # no header at all
            raw = bytearray(b"\\n".join(line for line in bytes(raw).split(b"\\n") if not line.startswith(b"#")))
        else:                                                    # huge numbers, empty fields
            raw = bytearray(bytes(raw).replace(b"\\t1|", b"\\t99999999999999999999|", 3).replace(b"\\tPASS\\t", b"\\t\\t", 2))
        text = raw.decode("latin-1")
        for flavour in ("Genome1000", "Falciparum"):
            flat = ha.FlatVcf(text, 2, flavour=flavour, quality_filter=bool(trial & 1))
            assert flat.V >= 0 and flat.G >= 0
        got = ha.InbreedInputs(texts[2] if trial % 2 else text, DATA_SOURCE["Gnomad2_1"], text)
        assert got.L >= 0


def test_inbreed_inputs_with_offsets_of_more_than_fourteen_alts():
    """A reference offset with more than 14 SNP alts (multi-base records that differ in one nucleotide) does not fit two 4-bit
    indices: every gt8 flattener -- whole text, a file in pieces, streamed -- leaves its cells as 16-bit wide rows (and its byte
    row 0xFF throughout), equal to the scaffold encoder's."""
    import tempfile
    from pathlib import Path

    from . import inbreed_inputs as ii
    from .records_io import DATA_SOURCE

    G, L = 23, 300
    rec, gt = sv.multiallelic_block(G, L, rng_seed=9, indel_frac=0.0, missing_af_frac=0.0, dup_records=0)
    rng = np.random.default_rng(2)
    offsets, refs, alts, afs = list(rec.offsets), list(rec.refs), [list(a) for a in rec.alts], [np.array(a) for a in rec.af]
    for at, n_alt in ((40, 19), (150, 15), (299, 21)):
        ref = "ACGTACG"
        al = [ref[:i] + b + ref[i + 1:] for i in range(7) for b in "ACGT" if b != ref[i]][:n_alt]
        refs[at], alts[at] = ref, al
        afs[at] = np.tile(rng.uniform(0.004, 0.03, n_alt).astype(np.float32).reshape(-1, 1), (1, 6))
        gt[at, :, 0] = np.where(rng.random(G) < 0.6, rng.integers(1, n_alt + 1, G), 0)
        gt[at, :, 1] = np.where(rng.random(G) < 0.6, rng.integers(1, n_alt + 1, G), 0)
    # a repeated record of a wide offset right behind it: same-phase pairs and three-variant cells in 16 bits
    offsets.insert(151, offsets[150]); refs.insert(151, refs[150]); alts.insert(151, list(alts[150])); afs.insert(151, afs[150].copy())
    gt = np.insert(gt, 151, gt[150], axis=0)
    gt[151, ::3, :] = 0
    rec = oa.Records(rec.contig, np.array(offsets, dtype=np.uint64), refs, alts, af=afs)
    for a in rec.af:
        a[:, 4] = a[:, 5]
    ids = [f"NA{i:05d}" for i in range(G)]
    ref_text = vt.write_vcf_mono(rec, "Gnomad2_1")
    dip_text = vt.write_vcf_1000(rec, gt, ids, rng_seed=1, quirks=False)
    loci = ii.ReferenceLoci(rec)
    want_bytes, want_wide, want_cells = ii.encode_wide(rec, gt, loci)
    assert len(want_wide) == 3 and (want_cells == 0xFFFF).any() and ((want_cells & 0xFF) == 0)[want_cells != 0].any()
    got = ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], dip_text, 2)
    assert got.error == "" and got.genome_ids == ids and got.amax == 30            # (the repeated record's alts join the offset's list)
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / "kg.vcf"
        path.write_text(dip_text)
        variants = [("text", got)]
        for chunk_bytes in (1, 4000, 0):
            variants.append((f"file {chunk_bytes}", ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=path, chunk_bytes=chunk_bytes)))
            variants.append((f"streamed {chunk_bytes}", ha.InbreedInputs(ref_text, DATA_SOURCE["Gnomad2_1"], None, 2, diploid_path=path, chunk_bytes=chunk_bytes, streaming=True)))
        for name, flat in variants:
            assert np.array_equal(flat.bytes, want_bytes), name
            assert np.array_equal(flat.wide_loci, want_wide) and np.array_equal(flat.wide_cells, want_cells), name
