"""The rsid / Ensembl indexes on columns (kgx_variant_sort.h) against the oracle's restatement of
VariantSort / SortedVariantAnalysis (oracle/kgo_sort.cpp) on the same VCF text: entry for entry, in index order."""
from __future__ import annotations

import numpy as np
import pytest

from . import host_api as H
from . import oracle_api as O

VEP_NAMES = ["Allele", "Consequence", "IMPACT", "SYMBOL", "Gene", "Feature_type"]
GENES = ["ENSG00000187634", "ENSG00000188976", "ENSG00000187961", "ENSG00000187583", "LRG_741", "ensg_lower", ""]


def sort_vcf(seed: int, flavour: str, n_records: int = 160, n_samples: int = 9, first_usable: bool = True, header: str = "plain") -> str:
    """VCF text with what the two indexes react to: identifiers that repeat, are missing (".") or end in white space;
    contigs out of order and offsets that repeat; vep entries of the right and of the wrong size, without a gene, bare or
    repeated vep keys; for the phased flavour, hets whose phase A allele is the higher alt."""
    rng = np.random.default_rng(seed)
    vep_description = {
        "plain": 'Description="Consequence annotations from Ensembl VEP. Format: ' + "|".join(VEP_NAMES) + '"',
        "no_gene": 'Description="Consequence annotations. Format: Allele|Consequence|IMPACT|SYMBOL|Feature_type"',
        "repeated": 'Description="Consequence annotations. Format: Allele|Gene|IMPACT|SYMBOL|Gene|Feature_type"',
        "equals": 'Description="Consequence annotations (a=b). Format: ' + "|".join(VEP_NAMES) + '"',
        "comma": 'Description="Consequence, annotations <from> VEP. Format: ' + "|".join(VEP_NAMES) + '"',
    }.get(header, "")
    lines = ["##fileformat=VCFv4.2",
             '##INFO=<ID=AF,Number=A,Type=Float,Description="Allele frequency, total">',
             "##INFO=<ID=vep,Number=.,Type=String," + vep_description + ">"]
    if header == "none":
        lines.pop()
    samples = [f"HG{int(x):05d}" for x in rng.permutation(n_samples) + 100]
    columns = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
    lines.append(columns + ("\tFORMAT\t" + "\t".join(samples) if flavour == "Genome1000" else ""))
    contigs = ["chr2", "chr10", "chr1"]
    position = {c: 1000 for c in contigs}
    opened = set()
    for r in range(n_records):
        contig = contigs[(r * 3 // n_records + int(rng.random() < 0.1)) % 3]
        if rng.random() > 0.15:
            position[contig] += int(rng.integers(1, 40))          # else: the offset repeats
        n_alt = int(rng.choice([1, 1, 1, 2, 3]))
        ref = "ACGT"[int(rng.integers(4))]
        alts = []
        for a in range(n_alt):
            alt = "ACGT"[(("ACGT".index(ref)) + 1 + a) % 4] + ("T" * int(rng.random() < 0.2))
            alts.append(alt)
        alt_field = ",".join(alts) if rng.random() > 0.03 else "."
        u = rng.random()
        if u < 0.5:
            ident = f"rs{1000 + r}"
        elif u < 0.7:
            ident = f"rs{1000 + int(rng.integers(0, max(r, 1)))}"         # an identifier seen before (maybe)
        elif u < 0.8:
            ident = f"rs{1000 + r} "                                       # trailing white space is trimmed
        else:
            ident = "."
        entries = []
        for _ in range(int(rng.choice([0, 1, 1, 2, 3]))):
            gene = GENES[int(rng.integers(len(GENES)))]
            fields = [alts[0], "missense_variant", "MODERATE", "SYM" + gene[-3:], gene, "Transcript"]
            shape = rng.random()
            if shape < 0.12:
                fields = fields[:5]                                        # one sub-field short
            elif shape < 0.2:
                fields = fields + ["HC", "extra"]                          # the Gnomad 3 LoF spill-over
            entries.append("|".join(fields))
        info = [f"AF={','.join(['0.1'] * n_alt)}"]
        opening = contig not in opened            # the first record of a contig: one of them is the first Variant visited
        opened.add(contig)
        if opening:
            entries = ["|".join([alts[0], "x", "LOW", "S", GENES[0], "Transcript"])] if first_usable else ["too|short"]
        style = 0.5 if opening else rng.random()
        if entries and style > 0.1:
            info.append("vep=" + ",".join(entries))
            if style > 0.9:
                info.append("vep=" + "|".join(["A", "c", "i", "s", "ENSG_SECOND_KEY", "t"]))    # the first key counts
        elif style <= 0.05:
            info.append("vep")                                             # a bare key holds no data
        rng.shuffle(info)
        fields = [contig, str(position[contig]), ident, ref, alt_field, "50", "PASS", ";".join(info)]
        if flavour == "Genome1000":
            n_listed = 1 if alt_field == "." else n_alt
            calls = []
            for _ in samples:
                if opening:
                    calls.append("1|0")                                   # every genome's first visit is its contig's opening record
                elif rng.random() < 0.7:
                    calls.append("0|0")
                else:
                    calls.append(f"{int(rng.integers(0, n_listed + 1))}|{int(rng.integers(0, n_listed + 1))}")
            fields += ["GT"] + calls
        lines.append("\t".join(fields))
    return "\n".join(lines) + "\n"


def oracle_population(text: str, flavour: str) -> O.Population:
    population = O.Population("sorted")
    if flavour == "MonoGenome":
        population.add_vcf_mono(text, "Gnomad2_1", "Reference")
    else:
        population.add_vcf_1000(text)
    return population


KINDS_ALL = ["ensembl", "non_ensembl", "allele_ensembl", "id", "genome_id"]


@pytest.mark.parametrize("flavour", ["MonoGenome", "Genome1000"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_indexes_match_oracle(flavour, seed):
    text = sort_vcf(seed, flavour)
    population = oracle_population(text, flavour)
    for what in KINDS_ALL:
        expect = population.variant_sort(what)
        got = H.variant_sort(text, flavour, what, threads=1 + seed)
        assert got == expect, what
        assert what == "non_ensembl" or len(expect) > 20, what
    assert population.variant_sort("genome_id_mt") == population.variant_sort("genome_id")
    # ensemblAddIndex with a gene list, and filterEnsembl (a code listed twice doubles its entries; unknown codes add nothing)
    listed = [GENES[1], GENES[4], "ENSG_NOT_THERE"]
    assert H.variant_sort(text, flavour, "ensembl", listed) == population.variant_sort("ensembl", listed)
    doubled = [GENES[0], GENES[2], GENES[0], "ENSG_NOT_THERE"]
    expect = population.variant_sort("filter", doubled)
    assert H.variant_sort(text, flavour, "filter", doubled) == expect
    assert len(expect) > 0
    assert int(population.variant_sort("non_ensembl")[0][0]) > 0


@pytest.mark.parametrize("flavour", ["MonoGenome", "Genome1000"])
def test_first_variant_decides_the_gene_column(flavour):
    """VariantSort looks the "Gene" column up on the first Variant it visits (kgl_variant_sort.cpp:56-63): when that one
    has no usable vep entry nothing is ever indexed."""
    # the first Variant visited is the first of the lowest contig, not the first of the file
    for seed in (4, 5):
        text = sort_vcf(seed, flavour, first_usable=False)
        population = oracle_population(text, flavour)
        expect = population.variant_sort("ensembl")
        assert H.variant_sort(text, flavour, "ensembl") == expect
    # force it: a file of one contig whose first record has only a mis-sized entry
    text = sort_vcf(6, flavour, first_usable=False)
    body = [line for line in text.split("\n") if line.startswith("#") or line.startswith("chr2\t")]
    first = next(i for i, line in enumerate(body) if not line.startswith("#"))
    if flavour == "Genome1000":       # someone must carry the first record for it to be visited
        cut = body[first].split("\t")
        cut[9:] = ["1|0"] * (len(cut) - 9)
        cut[4] = "G" if cut[3] != "G" else "A"
        body[first] = "\t".join(cut)
    text = "\n".join(body) + "\n"
    population = oracle_population(text, flavour)
    assert population.variant_sort("ensembl") == []
    assert H.variant_sort(text, flavour, "ensembl") == []
    assert H.variant_sort(text, flavour, "allele_ensembl") == []
    assert H.variant_sort(text, flavour, "id") == population.variant_sort("id") != []


@pytest.mark.parametrize("header", ["none", "no_gene", "repeated", "equals", "comma"])
def test_vep_header_shapes(header):
    """No vep header line, one without a Gene column, one naming a column twice (void), a description cut at '=' (the
    Format list is lost) and one with commas and angle brackets inside the quotes (intact)."""
    text = sort_vcf(7, "MonoGenome", header=header)
    population = oracle_population(text, "MonoGenome")
    expect = population.variant_sort("ensembl")
    assert H.variant_sort(text, "MonoGenome", "ensembl") == expect
    assert (len(expect) > 0) == (header == "comma")
    assert H.variant_sort(text, "MonoGenome", "allele_ensembl") == population.variant_sort("allele_ensembl")


def test_hand_worked_indexes():
    """Two genomes, one identifier on two records: the population index keeps the first genome's first visit, each
    genome's index its own; a het with the higher alt on phase A lists that alt first."""
    head = ["##fileformat=VCFv4.2",
            '##INFO=<ID=vep,Number=.,Type=String,Description="x Format: Allele|Gene">',
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tB\tA"]
    rows = ["1\t100\trsX\tA\tC,G\t.\tPASS\tvep=C|ENSG1,G|ENSG2,G|\tGT\t2|1\t0|0",
            "1\t200\trsX\tT\tC\t.\tPASS\tvep=C|ENSG1\tGT\t0|0\t0|1",
            "1\t300\t.\tT\tG\t.\tPASS\tvep=G|LRG_1\tGT\t1|1\t0|0"]
    text = "\n".join(head + rows) + "\n"
    # genome A (first by id) carries only the record at 200 -> rsX is its phase B variant there
    assert H.variant_sort(text, "Genome1000", "id") == [("rsX", "1:g.199T>C:2")]
    assert H.variant_sort(text, "Genome1000", "genome_id") == [("A", "rsX", "1:g.199T>C:2"), ("B", "rsX", "1:g.99A>G:1")]
    # visits: A: 199T>C:2 | B: 99A>G:1 (phase A first), 99A>C:2, 299T>G:1, 299T>G:2
    assert H.variant_sort(text, "Genome1000", "ensembl") == [
        ("ENSG1", "1:g.199T>C:2"), ("ENSG1", "1:g.99A>G:1"), ("ENSG1", "1:g.99A>C:2"),
        ("ENSG2", "1:g.99A>G:1"), ("ENSG2", "1:g.99A>C:2"),
        ("LRG_1", "1:g.299T>G:1"), ("LRG_1", "1:g.299T>G:2")]
    assert H.variant_sort(text, "Genome1000", "non_ensembl") == [("2",)]
    assert H.variant_sort(text, "Genome1000", "allele_ensembl") == [("rsX", "ENSG1,ENSG2")]
    population = oracle_population(text, "Genome1000")
    for what in KINDS_ALL:
        assert population.variant_sort(what) == H.variant_sort(text, "Genome1000", what), what


@pytest.mark.parametrize("flavour", ["MonoGenome", "Genome1000"])
def test_indexes_from_a_file_read_in_pieces(tmp_path, flavour):
    from . import vcf_text as vt

    text = sort_vcf(31, flavour)
    (tmp_path / "sites.vcf").write_text(text)
    (tmp_path / "sites.vcf.bgz").write_bytes(vt.bgzip(text.encode(), block=2000))
    for what in KINDS_ALL:
        want = H.variant_sort(text, flavour, what)
        for name in ("sites.vcf", "sites.vcf.bgz"):
            for chunk_bytes in (1, 900, 0):
                assert H.variant_sort(None, flavour, what, path=tmp_path / name, chunk_bytes=chunk_bytes, threads=2) == want, (what, name, chunk_bytes)
    with pytest.raises(IOError):
        H.variant_sort(None, flavour, "id", path=tmp_path / "missing.vcf")


def test_empty_inputs():
    for flavour in ("MonoGenome", "Genome1000"):
        for text in ("", "##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n", "1\t5\n"):
            for what in ("ensembl", "id", "genome_id", "allele_ensembl"):
                assert H.variant_sort(text, flavour, what) == []
            assert H.variant_sort(text, flavour, "non_ensembl") == [("0",)]
