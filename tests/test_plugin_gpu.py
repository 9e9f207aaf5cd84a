"""The C++ analysis package GpuAlleleAnalysis driven through the VirtualAnalysis surface
(initializeAnalysis -> fileReadAnalysis -> iterationAnalysis -> finalizeAnalysis) by kgx_host_driver,
its CSV output compared with the oracle's CalcFWS / HeteroHomoZygous restatement.  Needs a GPU."""
import numpy as np
import pytest

from . import oracle_api as oa
from . import records_io as rio
from . import synth_vcf as sv

pytestmark = pytest.mark.gpu

# "StartSeed" of GPU_INBREED = the oracle's start_seed: the k-th per-genome task owns std::mt19937_64(seed + k)
START_SEED = 4242


# Device bindings of the package (kgx_device_binding.h): the default single device; "Devices=0" = every visible device
# (one on the test box); "DeviceList=0,0,0" = three genome shards, here on one device -- the single-process multi-GPU
# path (one host thread per shard, the count exchange) with the same CSVs byte for byte.
BINDINGS = [{}, {"Devices": 0}, {"DeviceList": "0,0,0"}]


@pytest.mark.parametrize("binding", BINDINGS, ids=["one-device", "all-visible", "three-shards"])
@pytest.mark.parametrize("mode,source", [(oa.Population.UNPHASED, "Falciparum"), (oa.Population.PHASED, "Genome1000")])
def test_gpu_allele_package_matches_oracle(tmp_path, mode, source, binding, kgx):
    G, L = 153, 1500
    rec, gt = sv.multiallelic_block(G, L, rng_seed=21 + mode)
    ids = sv.genome_ids(G, prefix="PF")
    path = tmp_path / "pop.bin"
    rio.write_records(path, rec, gt, ids, mode, source, population_id="Pf7")
    res = rio.run_driver("GPU_ALLELE", tmp_path, [path], **binding)
    assert res.returncode == 0, res.stderr

    opop = sv.oracle_population(rec, gt, ids, mode)
    variant_out, genome_out, vdb = opop.fws()
    hgvs = [vdb.hgvs(i) for i in range(vdb.n_variants)]

    # VariantFWS.csv: one line per distinct variant in lexicographic HGVS order, counts bit-exact
    header, rows = rio.read_csv(tmp_path / "VariantFWS.csv")
    assert header[0] == "Variant" and header[-3:] == ["Hom Ref (A;A)", "Het Ref Minor (A;a)", "Hom Minor (a;a)"]
    assert [r[0] for r in rows] == hgvs
    got = np.array([[int(x) for x in r[-3:]] for r in rows], dtype=np.uint64)
    assert np.array_equal(got, variant_out)

    # GenomeFWS.csv: genome-id order, 11 bins x (ref, het, hom)
    header, rows = rio.read_csv(tmp_path / "GenomeFWS.csv")
    assert [r[0] for r in rows] == sorted(ids)
    got = np.array([[int(r[1 + 8 * b + 5 + k]) for b in range(11) for k in range(3)] for r in rows], dtype=np.uint64)
    assert np.array_equal(got.reshape(len(rows), 11, 3), genome_out)
    lows = [float(rows[0][1 + 8 * b]) for b in range(11)]
    assert lows == [0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5]

    # VariantStatistics.csv: HeteroHomoZygous counters per genome x contig (incl. compound offsets and > 2 copies)
    header, rows = rio.read_csv(tmp_path / "VariantStatistics.csv")
    want = opop.hethom(rec.contig)     # total,snp,indel,hom_minor,het_minor,het_ref_minor,hom_ref
    got = np.array([[int(r[2]), int(r[3]), int(r[4]), int(r[7]), int(r[8]), int(r[6]), int(r[5])] for r in rows], dtype=np.uint64)
    assert [r[0] for r in rows] == sorted(ids) and all(r[1] == rec.contig for r in rows)
    assert np.array_equal(got, want)
    assert want[:, 4].sum() > 0 and want[:, 2].sum() > 0      # compound offsets and indels were exercised
    # Wright's F_IS column against the population aggregate
    agg = want.sum(0)
    for r, w in zip(rows, want):
        fis = oa.lib().kgo_wrights_fis(oa._p(np.ascontiguousarray(agg)), oa._p(np.ascontiguousarray(w)))
        assert float(r[9]) == pytest.approx(fis, rel=1e-5, abs=1e-9)      # CSV prints 6 significant digits


def test_gpu_allele_package_disables_itself_on_bad_device(tmp_path, kgx):
    rec, gt = sv.multiallelic_block(4, 10)
    path = tmp_path / "pop.bin"
    rio.write_records(path, rec, gt, sv.genome_ids(4), oa.Population.UNPHASED, "Falciparum")
    res = rio.run_driver("GPU_ALLELE", tmp_path, [path], Device=99)
    assert res.returncode == 1 and "initializeAnalysis failed" in res.stderr
    # a malformed device list disables the package the same way (no exception out of initializeAnalysis)
    for bad in ("0,a", "x", "0,-1", "99999999999999999999"):
        res = rio.run_driver("GPU_ALLELE", tmp_path, [path], DeviceList=bad)
        assert res.returncode == 1 and "initializeAnalysis failed" in res.stderr, (bad, res.returncode, res.stderr[-300:])


@pytest.mark.parametrize("algorithm,mode,source,binding", [("Simple", oa.Population.PHASED, "Genome1000", {}),
                                                           ("RitlandLocus", oa.Population.UNPHASED, "Falciparum", {}),
                                                           ("HallME", oa.Population.PHASED, "Genome1000", {}),
                                                           ("Loglikelihood", oa.Population.PHASED, "Genome1000", {}),
                                                           ("Simple", oa.Population.PHASED, "Genome1000", {"DeviceList": "0,0"}),
                                                           ("Loglikelihood", oa.Population.PHASED, "Genome1000", {"DeviceList": "0,0,0"})])
def test_gpu_inbreed_package_matches_oracle_window_loop(tmp_path, kgx, algorithm, mode, source, binding):
    """GPU_INBREED through the VirtualAnalysis surface vs the oracle's populationInbreeding window loop."""
    G, L = 267, 2000
    rec, gt = sv.multiallelic_block(G, L, rng_seed=31, missing_af_frac=0.03, dup_records=50)
    for a in rec.af:                       # Gnomad 2.1 reads SAS from the same "AF" field as ALL (kgl_variant_db_freq.h:92)
        a[:, 4] = a[:, 5]
    ids = sv.genome_ids(G, prefix="NA")
    pops = ["AFR", "AMR", "EAS", "EUR", "SAS"]
    ped = [(g, pops[i % 5]) for i, g in enumerate(ids) if i % 13 != 7]      # a few genomes have no PED record
    ref_path, dip_path = tmp_path / "gnomad.bin", tmp_path / "diploid.bin"
    rio.write_records(ref_path, rec, None, ["Reference"], oa.Population.REFERENCE, "Gnomad2_1", population_id="Gnomad")
    rio.write_records(dip_path, rec, gt, ids, mode, source, population_id="Diploid", ped=ped)
    params = dict(AnalysisType="false", OutputFile="inbreed", Algorithm=algorithm, MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                  LowerWindow=0, UpperWindow=60000, LociiCount=150, SamplingDistance=40, StartSeed=START_SEED)
    res = rio.run_driver("GPU_INBREED", tmp_path, [ref_path, dip_path], **params, **binding)
    assert res.returncode == 0, res.stderr

    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    dip = sv.oracle_population(rec, gt, ids, mode)
    ped_map = dict(ped)
    sorted_ids = sorted(ids)
    sp_of = np.array([oa.SUPER_POPS.index(ped_map[g]) if g in ped_map else -1 for g in sorted_ids], dtype=np.int32)
    cols = oa.population_inbreeding(ref.filter_snp_pass(), dip, sp_of, algorithm, 0, 60000, 40, 150, 0.02, 0.9, seed=START_SEED)
    assert len(cols) >= 3

    header, rows = rio.read_csv(tmp_path / "inbreed_detail.csv")
    got = {}
    for r in rows:
        got[(r[0], r[1])] = ([int(r[2]), int(r[4]), int(r[6]), int(r[8]), int(r[10])], [float(r[3]), float(r[5]), float(r[7]), float(r[9]), float(r[11])])
    n_checked = 0
    for ident, counts, freqs, present in cols:
        for k, g in enumerate(sorted_ids):
            if not present[k]:
                assert (ident, g) not in got
                continue
            c, f = got[(ident, g)]
            assert c == counts[k].tolist(), (ident, g)                       # major_het, minor_het, minor_hom, major_hom, total
            assert np.allclose(f[:4], freqs[k, :4], rtol=1e-12, atol=1e-12)
            tol = {"Simple": 1e-10, "RitlandLocus": 1e-10, "HallME": 1e-9, "Loglikelihood": 2e-6}[algorithm]
            assert abs(f[4] - freqs[k, 4]) <= tol, (ident, g, f[4], freqs[k, 4])
            n_checked += 1
    assert n_checked == len(got) and n_checked >= 3 * (G - 21)
    # the summary CSV: header line, one column per window with the reference's ident contig_lower_upper
    lines = (tmp_path / "inbreed.csv").read_text().strip().split("\n")
    assert lines[0].startswith("DriverParameters,Algorithm:" + algorithm)
    assert lines[1].split(",")[:9] == ["Sample", "Population", "Description", "SuperPopulation", "Description", "Relationship", "Sex", "Mother", "Father"]
    assert lines[1].split(",")[9:] == [c[0] for c in cols]
    assert len(lines) == 2 + int(np.sum(sp_of >= 0))


@pytest.mark.parametrize("path", ["streamed", "two-phase"])
def test_gpu_allele_package_reads_vcf_directly(tmp_path, kgx, path):
    """A FileNameOnly data file: the package parses the VCF itself (no Variant / PopulationDB objects) and must
    produce what the oracle gets by parsing the same text the reference's way and running CalcFWS / HeteroHomoZygous.
    "streamed": rows go to the device piece by piece while the file is read (file order, repeated records merged at the
    end); "two-phase": a sample that carries no variant is no genome, which the streaming flattener cannot know in time
    -- the package says so and takes the two-phase flattener."""
    from . import vcf_text as vt

    G, L = 37, 1200
    rec, gt = sv.multiallelic_block(G, L, rng_seed=9, dup_records=2)
    if path == "two-phase":
        gt[:, 5, :] = 0
    ids = [f"HG{i:05d}" for i in range(G)]
    text = vt.write_vcf_1000(rec, gt, ids, rng_seed=4, quirks=(path == "streamed"))
    vcf = tmp_path / "pop.vcf"
    vcf.write_text(text)
    res = rio.run_driver("GPU_ALLELE", tmp_path, [f"vcf:{vcf}"])
    assert res.returncode == 0, res.stderr
    assert ("flattened in two phases" in res.stderr) == (path == "two-phase"), res.stderr[-2000:]

    opop = oa.Population("vcf")
    opop.add_vcf_1000(text)
    variant_out, genome_out, vdb = opop.fws()
    header, rows = rio.read_csv(tmp_path / "VariantFWS.csv")
    assert [r[0] for r in rows] == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert np.array_equal(np.array([[int(x) for x in r[-3:]] for r in rows], dtype=np.uint64), variant_out)
    header, rows = rio.read_csv(tmp_path / "GenomeFWS.csv")
    assert [r[0] for r in rows] == [vdb.genome_id(i) for i in range(vdb.n_genomes)]
    got = np.array([[int(r[1 + 8 * b + 5 + k]) for b in range(11) for k in range(3)] for r in rows], dtype=np.uint64)
    assert np.array_equal(got.reshape(len(rows), 11, 3), genome_out)
    header, rows = rio.read_csv(tmp_path / "VariantStatistics.csv")
    want = opop.hethom(rec.contig)
    got = np.array([[int(r[2]), int(r[3]), int(r[4]), int(r[7]), int(r[8]), int(r[6]), int(r[5])] for r in rows], dtype=np.uint64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("algorithm", ["Simple", "Loglikelihood"])
def test_gpu_inbreed_package_synthetic_self_check(tmp_path, kgx, algorithm):
    """AnalysisType=true: SyntheticAnalysis::syntheticInbreeding (kga_analysis_inbreed_synthetic.cpp:17-138) with the
    synthetic genomes drawn on the device.  Only the reference (Gnomad) file is needed; the CSV has the layout of
    InbreedingOutput::writeSynthetic and the estimates track the F each genome id encodes."""
    L = 6000
    rng = np.random.default_rng(3)
    offsets = np.arange(1, L + 1, dtype=np.uint64) * 50
    af = rng.uniform(0.05, 0.5, L).astype(np.float32)
    rec = oa.Records("chr7", offsets, ["A"] * L, [["T"]] * L, af=[np.tile(np.float32(a), (1, 6)) for a in af])
    ref_path = tmp_path / "gnomad.bin"
    rio.write_records(ref_path, rec, None, ["Reference"], oa.Population.REFERENCE, "Gnomad2_1", population_id="Gnomad")
    params = dict(AnalysisType="true", OutputFile="syn", Algorithm=algorithm, MinAlleleFreq=0.0, MaxAlleleFreq=1.0,
                  LowerWindow=0, UpperWindow=L * 50 + 10, LociiCount=2000, SamplingDistance=50, SyntheticSeed=99)
    res = rio.run_driver("GPU_INBREED", tmp_path, [ref_path], **params)
    assert res.returncode == 0, res.stderr
    lines = (tmp_path / "syn.csv").read_text().strip().split("\n")
    assert lines[0].startswith("DriverParameters,Algorithm:" + algorithm)
    assert lines[1] == "Sample,SynInbreed,CalcInbreed"
    rows = [ln.split(",") for ln in lines[2:]]
    # 101 genomes per super population (kgl_variant_db_freq.h:64-69 lists six, ALL included), ids as
    # generateSyntheticGenomeId writes them (_syngen.cpp:202-222)
    assert len(rows) == 101 * 6
    by_sp = {}
    for r in rows:
        sp, code, counter = r[0].split("_")
        f = -int(code[1:]) / 1e6 if code.startswith("N") else int(code) / 1e6
        assert abs(float(r[1]) - f) < 1e-9                      # generateInbreeding decodes what the id encodes
        assert abs(f - (-0.5 + 0.01 * int(counter))) < 2e-6
        by_sp.setdefault(sp, []).append((f, [float(x) for x in r[2:] if x != ""]))
    assert sorted(by_sp) == ["AFR", "ALL", "AMR", "EAS", "EUR", "SAS"]
    for sp, vals in by_sp.items():
        syn = np.array([v[0] for v in vals])
        calc = np.array([v[1] for v in vals])
        assert calc.shape[1] == 3        # one CalcInbreed per 2000-locus window of the 6000 loci
        for c in range(calc.shape[1]):
            slope, intercept = np.polyfit(syn, calc[:, c], 1)
            assert slope > 0.8 and abs(intercept) < 0.05, (sp, c, slope, intercept)
    # the two windows use different draws
    assert not np.array_equal(calc[:, 0], calc[:, 1])


@pytest.mark.parametrize("quality_filter", [False, True])
def test_gpu_allele_package_reads_pf_vcf(tmp_path, kgx, quality_filter):
    """VcfFlavour=Falciparum: the package parses the unphased Pf7-style VCF itself (PfVCFImpl's rules, canonical
    variants, every sample a genome holding every header contig), optionally drops the records P7VariantFilter rejects,
    and must produce what the oracle gets from the same text through Variant objects, viewFilter and CalcFWS /
    HeteroHomoZygous."""
    from . import vcf_text as vt

    G, L = 45, 2500
    ids = [f"PF{i:04d}-C" for i in range(G)]
    text = vt.write_vcf_pf(L, ids, rng_seed=21)
    vcf = tmp_path / "pf.vcf"
    vcf.write_text(text)
    res = rio.run_driver("GPU_ALLELE", tmp_path, [f"vcf:{vcf}"], VcfFlavour="Falciparum",
                         Pf7QualityFilter="TRUE" if quality_filter else "FALSE")
    assert res.returncode == 0, res.stderr

    opop = oa.Population("pf")
    opop.add_vcf_pf(text)
    if quality_filter:
        opop = opop.filter_p7()
    variant_out, genome_out, vdb = opop.fws()
    header, rows = rio.read_csv(tmp_path / "VariantFWS.csv")
    assert [r[0] for r in rows] == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert np.array_equal(np.array([[int(x) for x in r[-3:]] for r in rows], dtype=np.uint64), variant_out)
    header, rows = rio.read_csv(tmp_path / "GenomeFWS.csv")
    assert [r[0] for r in rows] == [vdb.genome_id(i) for i in range(vdb.n_genomes)] == sorted(ids)
    got = np.array([[int(r[1 + 8 * b + 5 + k]) for b in range(11) for k in range(3)] for r in rows], dtype=np.uint64)
    assert np.array_equal(got.reshape(len(rows), 11, 3), genome_out)
    header, rows = rio.read_csv(tmp_path / "VariantStatistics.csv")
    got = {(r[0], r[1]): [int(r[2]), int(r[3]), int(r[4]), int(r[7]), int(r[8]), int(r[6]), int(r[5])] for r in rows}
    n_rows = 0
    for contig in ["Pf3D7_01_v3", "Pf3D7_02_v3", "Pf3D7_MIT_v3", "Pf3D7_API_v3"]:     # the last one has no record at all
        want, present = opop.hethom(contig), opop.hethom_present(contig)
        for g, genome in enumerate(sorted(ids)):
            assert ((genome, contig) in got) == bool(present[g]), (genome, contig)
            if present[g]:
                assert got[(genome, contig)] == want[g].tolist(), (genome, contig)
                n_rows += 1
    assert n_rows == len(got)
    if not quality_filter:
        assert n_rows == 4 * G                                     # every genome holds every header contig


def test_gpu_allele_package_sweeps_more_than_63_contigs(tmp_path, kgx):
    """One by-genome sweep holds 63 contigs' worth of bins; a population with more (here 150: three sweeps) is swept once
    per 63 -- HeteroHomoZygous per genome x contig as the oracle has it."""
    from . import vcf_text as vt

    G, L = 23, 4000
    ids = [f"PF{i:04d}-C" for i in range(G)]
    contigs = [f"Pf_scaffold_{i:03d}" for i in range(150)] + ["Pf_unused"]
    text = vt.write_vcf_pf(L, ids, rng_seed=33, contigs=contigs)
    vcf = tmp_path / "pf.vcf"
    vcf.write_text(text)
    res = rio.run_driver("GPU_ALLELE", tmp_path, [f"vcf:{vcf}"], VcfFlavour="Falciparum", Pf7QualityFilter="FALSE")
    assert res.returncode == 0, res.stderr
    opop = oa.Population("pf")
    opop.add_vcf_pf(text)
    header, rows = rio.read_csv(tmp_path / "VariantStatistics.csv")
    got = {(r[0], r[1]): [int(r[2]), int(r[3]), int(r[4]), int(r[7]), int(r[8]), int(r[6]), int(r[5])] for r in rows}
    n_rows = 0
    for contig in contigs:
        want, present = opop.hethom(contig), opop.hethom_present(contig)
        for g, genome in enumerate(sorted(ids)):
            assert ((genome, contig) in got) == bool(present[g]), (genome, contig)
            if present[g]:
                assert got[(genome, contig)] == want[g].tolist(), (genome, contig)
                n_rows += 1
    assert n_rows == len(got) == len(contigs) * G


@pytest.mark.parametrize("algorithm", ["Simple", "HallME"])
def test_gpu_inbreed_package_reads_vcf_directly(tmp_path, kgx, algorithm):
    """Both INBREED inputs as "FileNameOnly" VCF files: the package flattens the Gnomad site file into its reference
    contig and the 1000-Genomes VCF into allele-index bytes itself.  Same window loop, same CSVs as from PopulationDB
    objects -- checked against the oracle fed by its own restatements of the two parsers."""
    from . import vcf_text as vt

    G, L = 53, 1800
    rec, gt = sv.multiallelic_block(G, L, rng_seed=17, missing_af_frac=0.03, dup_records=50)
    for a in rec.af:
        a[:, 4] = a[:, 5]                   # Gnomad 2.1 reads SAS from the same "AF" field as ALL
    ids = sv.genome_ids(G, prefix="HG")
    pops = ["AFR", "AMR", "EAS", "EUR", "SAS"]
    ped = [(g, pops[i % 5]) for i, g in enumerate(ids) if i % 11 != 3]
    ref_text = vt.write_vcf_mono(rec, "Gnomad2_1")
    dip_text = vt.write_vcf_1000(rec, gt, ids, rng_seed=8)
    (tmp_path / "gnomad.vcf").write_text(ref_text)
    (tmp_path / "kg.vcf").write_text(dip_text)
    (tmp_path / "ped.txt").write_text("".join(f"{g}\t{sp}\n" for g, sp in ped))
    params = dict(AnalysisType="false", OutputFile="inbreed", Algorithm=algorithm, MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                  LowerWindow=0, UpperWindow=60000, LociiCount=150, SamplingDistance=40, StartSeed=START_SEED)
    res = rio.run_driver("GPU_INBREED", tmp_path, [f"vcf:Gnomad2_1:{tmp_path / 'gnomad.vcf'}", f"vcf:Genome1000:{tmp_path / 'kg.vcf'}",
                                                    f"ped:{tmp_path / 'ped.txt'}"], **params)
    assert res.returncode == 0, res.stderr

    ref = oa.Population("gnomad")
    ref.add_vcf_mono(ref_text, "Gnomad2_1")
    dip = oa.Population("kg")
    dip.add_vcf_1000(dip_text)
    vdb_ids = [oa.VariantDB(dip).genome_id(i) for i in range(dip.genome_count())]
    ped_map = dict(ped)
    sp_of = np.array([oa.SUPER_POPS.index(ped_map[g]) if g in ped_map else -1 for g in vdb_ids], dtype=np.int32)
    cols = oa.population_inbreeding(ref.filter_snp_pass(), dip, sp_of, algorithm, 0, 60000, 40, 150, 0.02, 0.9, seed=START_SEED)
    assert len(cols) >= 3
    header, rows = rio.read_csv(tmp_path / "inbreed_detail.csv")
    got = {(r[0], r[1]): ([int(r[2]), int(r[4]), int(r[6]), int(r[8]), int(r[10])], [float(r[3]), float(r[5]), float(r[7]), float(r[9]), float(r[11])])
           for r in rows}
    n_checked = 0
    for ident, counts, freqs, present in cols:
        for k, g in enumerate(vdb_ids):
            if not present[k]:
                assert (ident, g) not in got
                continue
            c, f = got[(ident, g)]
            assert c == counts[k].tolist(), (ident, g)
            assert np.allclose(f[:4], freqs[k, :4], rtol=1e-12, atol=1e-12)
            assert abs(f[4] - freqs[k, 4]) <= {"Simple": 1e-10, "HallME": 1e-9}[algorithm], (ident, g, f[4], freqs[k, 4])
            n_checked += 1
    assert n_checked == len(got) and n_checked >= 3 * (G - 10)


@pytest.mark.parametrize("via", ["vcf", "records"])
@pytest.mark.parametrize("algorithm", ["Simple", "Loglikelihood"])
def test_gpu_inbreed_package_takes_offsets_with_more_than_14_alts(tmp_path, kgx, via, algorithm):
    """Offsets with 16 and 20 same-length ("SNP") alts -- of the 16, two without a frequency for any super population: the
    reference's AlleleFreqVector has no cap (kga_analysis_inbreed_freq.cpp:18-57).  The matrix bytes' 4-bit indices hold 14, so
    the package keeps such offsets' cells as 16-bit WIDE ROWS (kgx_gt8_set_wide_rows), gives the windows that sample them a
    frequency table as wide as the widest and sweeps those windows on their own -- with the oracle's results, counts bit for
    bit, nothing cut, through both entries (VCF text streamed to the device; the parsers' PopulationDB objects)."""
    from . import vcf_text as vt

    G, L = 41, 900
    rec, gt = sv.multiallelic_block(G, L, rng_seed=23, indel_frac=0.0, missing_af_frac=0.0, dup_records=0)
    rng = np.random.default_rng(5)
    offsets, refs, alts, afs = list(rec.offsets), list(rec.refs), [list(a) for a in rec.alts], [np.array(a) for a in rec.af]
    wide_gt = []
    for k in range(6):                                             # six wide offsets past the block
        # isSNP: same length, one position differs -- a 7-mer has 21 such alts; 16 or 20 of them, in a shuffled order
        ref = "".join("ACGT"[int(i)] for i in rng.integers(0, 4, 7))
        al = [ref[:i] + b + ref[i + 1:] for i in range(7) for b in "ACGT" if b != ref[i]]
        al = [al[int(i)] for i in rng.permutation(len(al))[:(16 if k % 2 == 0 else 20)]]
        p = rng.uniform(0.005, 0.04, len(al))
        af = np.tile(p.astype(np.float32).reshape(-1, 1), (1, 6))
        if k % 2 == 0:
            af[[2, 9], :] = np.nan                                   # two alts nobody has a frequency for
        offsets.append(int(offsets[L - 1]) + 100 * (k + 1)); refs.append(ref); alts.append(al); afs.append(af)
        probs = np.concatenate([[0.5], np.full(len(al), 0.5 / len(al))])
        wide_gt.append(rng.choice(len(al) + 1, size=(G, 2), p=probs).astype(np.uint8))
    rec = oa.Records(rec.contig, np.array(offsets, dtype=np.uint64), refs, alts, af=afs)
    gt = np.concatenate([gt, np.stack(wide_gt)])
    for a in rec.af:
        a[:, 4] = a[:, 5]
    ids = sv.genome_ids(G, prefix="HG")
    pops = ["AFR", "AMR", "EAS", "EUR", "SAS"]
    ped = [(g, pops[i % 5]) for i, g in enumerate(ids)]
    upper = int(offsets[-1]) + 10
    # several windows: the last ones sample the wide offsets, the first ones do not (they go to the device as a batch)
    params = dict(AnalysisType="false", OutputFile="inbreed", Algorithm=algorithm, MinAlleleFreq=0.0, MaxAlleleFreq=1.0,
                  LowerWindow=0, UpperWindow=upper, LociiCount=250, SamplingDistance=1, StartSeed=START_SEED)
    if via == "vcf":
        ref_text = vt.write_vcf_mono(rec, "Gnomad2_1")
        dip_text = vt.write_vcf_1000(rec, gt, ids, rng_seed=8)
        (tmp_path / "gnomad.vcf").write_text(ref_text)
        (tmp_path / "kg.vcf").write_text(dip_text)
        (tmp_path / "ped.txt").write_text("".join(f"{g}\t{sp}\n" for g, sp in ped))
        res = rio.run_driver("GPU_INBREED", tmp_path, [f"vcf:Gnomad2_1:{tmp_path / 'gnomad.vcf'}", f"vcf:Genome1000:{tmp_path / 'kg.vcf'}",
                                                        f"ped:{tmp_path / 'ped.txt'}"], **params)
        ref = oa.Population("gnomad")
        ref.add_vcf_mono(ref_text, "Gnomad2_1")
        dip = oa.Population("kg")
        dip.add_vcf_1000(dip_text)
    else:
        ref_path, dip_path = tmp_path / "gnomad.bin", tmp_path / "diploid.bin"
        rio.write_records(ref_path, rec, None, ["Reference"], oa.Population.REFERENCE, "Gnomad2_1", population_id="Gnomad")
        rio.write_records(dip_path, rec, gt, ids, oa.Population.PHASED, "Genome1000", population_id="Diploid", ped=ped)
        res = rio.run_driver("GPU_INBREED", tmp_path, [ref_path, dip_path], **params)
        ref = oa.Population("gnomad")
        ref.add_genomes(["Reference"])
        ref.add_records(rec, None, oa.Population.REFERENCE)
        dip = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    assert res.returncode == 0, res.stderr
    log = res.stdout + res.stderr
    assert "6 reference offsets hold more than 14 SNP alts (the widest 20)" in log and "lost a frequency-bearing alt" not in log, log[-600:]

    vdb_ids = [oa.VariantDB(dip).genome_id(i) for i in range(dip.genome_count())]
    ped_map = dict(ped)
    sp_of = np.array([oa.SUPER_POPS.index(ped_map[g]) for g in vdb_ids], dtype=np.int32)
    cols = oa.population_inbreeding(ref.filter_snp_pass(), dip, sp_of, algorithm, 0, upper, 1, 250, 0.0, 1.0, seed=START_SEED)
    assert len(cols) >= 3
    header, rows = rio.read_csv(tmp_path / "inbreed_detail.csv")
    got = {(r[0], r[1]): ([int(r[2]), int(r[4]), int(r[6]), int(r[8]), int(r[10])], [float(r[3]), float(r[5]), float(r[7]), float(r[9]), float(r[11])])
           for r in rows}
    n_checked = 0
    tol = {"Simple": 1e-10, "Loglikelihood": 2e-6}[algorithm]
    for ident, counts, freqs, present in cols:
        for k, g in enumerate(vdb_ids):
            if not present[k]:
                continue
            c, f = got[(ident, g)]
            assert c == counts[k].tolist(), (ident, g, c, counts[k].tolist())
            assert np.allclose(f[:4], freqs[k, :4], rtol=1e-12, atol=1e-12)
            assert abs(f[4] - freqs[k, 4]) <= tol, (ident, g, f[4], freqs[k, 4])
            n_checked += 1
    assert n_checked == len(got) and n_checked >= 3 * G


@pytest.mark.parametrize("via", ["records", "vcf"])
def test_gpu_allele_package_bins_each_copy_by_its_own_record(tmp_path, kgx, via):
    """Repeated records of one variant with DIFFERENT AF: CalcFWS puts each Variant object into the bin of its own
    record.  Both flatteners (PopulationDB objects, VCF text) emit per-bin split rows for such variants; the by-genome
    bin counts must equal the oracle's, and the per-variant counts still merge every copy."""
    from . import vcf_text as vt

    G, L = 31, 900
    rec, gt = sv.multiallelic_block(G, L, rng_seed=77, dup_records=60)
    rng = np.random.default_rng(9)
    for a in rec.af:
        a[:, 5] = rng.uniform(0, 0.6, a.shape[0]).astype(np.float32)      # every record its own AF
    ids = sv.genome_ids(G, prefix="HG")
    if via == "records":
        path = tmp_path / "pop.bin"
        rio.write_records(path, rec, gt, ids, oa.Population.PHASED, "Genome1000", population_id="kg")
        res = rio.run_driver("GPU_ALLELE", tmp_path, [path])
        opop = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    else:
        text = vt.write_vcf_1000(rec, gt, ids, rng_seed=2, quirks=False)
        (tmp_path / "kg.vcf").write_text(text)
        res = rio.run_driver("GPU_ALLELE", tmp_path, [f"vcf:{tmp_path / 'kg.vcf'}"])
        opop = oa.Population("kg")
        opop.add_vcf_1000(text)
    assert res.returncode == 0, res.stderr
    variant_out, genome_out, vdb = opop.fws()
    header, rows = rio.read_csv(tmp_path / "VariantFWS.csv")
    assert [r[0] for r in rows] == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert np.array_equal(np.array([[int(x) for x in r[-3:]] for r in rows], dtype=np.uint64), variant_out)
    header, rows = rio.read_csv(tmp_path / "GenomeFWS.csv")
    got = np.array([[int(r[1 + 8 * b + 5 + k]) for b in range(11) for k in range(3)] for r in rows], dtype=np.uint64)
    assert np.array_equal(got.reshape(len(rows), 11, 3), genome_out)
    header, rows = rio.read_csv(tmp_path / "VariantStatistics.csv")
    want = opop.hethom(rec.contig)
    got = np.array([[int(r[2]), int(r[3]), int(r[4]), int(r[7]), int(r[8]), int(r[6]), int(r[5])] for r in rows], dtype=np.uint64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("algorithm", ["Simple", "RitlandLocus", "HallME", "Loglikelihood"])
def test_gpu_inbreed_package_writes_the_reference_ped_file(tmp_path, kgx, algorithm):
    """the result file of INBREED.  The package reads the reference's own PED file format (header +
    15 tab-separated fields, ParseHsGenomeGenealogyFile) and writes InbreedingOutput::writePedResults' layout; the oracle
    writes the same file from its own window loop with the reference's writer restated.  Byte for byte equal wherever the
    coefficients agree past the six digits the reference prints (Simple, RitlandLocus, HallME); Loglikelihood's 1e-5 band
    is compared field by field."""
    G, L = 120, 2000
    rec, gt = sv.multiallelic_block(G, L, rng_seed=33, missing_af_frac=0.02, dup_records=20)
    for a in rec.af:
        a[:, 4] = a[:, 5]
    ids = sv.genome_ids(G, prefix="NA")
    pops = [("ACB", "African Caribbean in Barbados", "AFR", "African"), ("MXL", "Mexican Ancestry in Los Angeles", "AMR", "American"),
            ("CHB", "Han Chinese in Beijing", "EAS", "East Asian"), ("GBR", "British in England and Scotland", "EUR", "European"),
            ("PJL", "Punjabi in Lahore", "SAS", "South Asian")]
    ped_rows, ped_lines = [], ["Family ID\tIndividual ID\tPaternal ID\tMaternal ID\tGender\tPhenotype\tPopulation\tPopulation Description\t"
                               "Super Population\tSuper Description\tRelationship\tSiblings\tSecond Order\tThird Order\tOther Comments"]
    for i, g in enumerate(ids):
        if i % 17 == 5:
            continue                                          # no PED record: the genome is skipped by the sweep and the writer
        pop, pop_desc, sp, sp_desc = pops[i % 5]
        father, mother = (ids[i - 1], ids[i - 2]) if i % 9 == 8 else ("0", "0")
        relationship = "child" if i % 9 == 8 else "unrel"
        sex = "1" if i % 2 else "2"
        ped_lines.append("\t".join([f"FAM{i // 3}", g, father, mother, sex, "0", pop, pop_desc, sp, sp_desc, relationship, "0", "0", "0", "0"]))
        ped_rows.append([g, pop, pop_desc, sp, sp_desc, relationship, sex, mother, father])
    ped_path = tmp_path / "samples.ped"
    ped_path.write_text("\n".join(ped_lines) + "\n")
    ref_path, dip_path = tmp_path / "gnomad.bin", tmp_path / "diploid.bin"
    rio.write_records(ref_path, rec, None, ["Reference"], oa.Population.REFERENCE, "Gnomad2_1", population_id="Gnomad")
    rio.write_records(dip_path, rec, gt, ids, oa.Population.PHASED, "Genome1000", population_id="Diploid")
    params = dict(AnalysisType="false", OutputFile="inbreed", Algorithm=algorithm, MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                  LowerWindow=0, UpperWindow=60000, LociiCount=150, SamplingDistance=40, StartSeed=START_SEED)
    res = rio.run_driver("GPU_INBREED", tmp_path, [f"ped:{ped_path}", ref_path, dip_path], **params)
    assert res.returncode == 0, res.stderr

    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    dip = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    sp_of_genome = {row[0]: oa.SUPER_POPS.index(row[3]) for row in ped_rows}
    sp_of = np.array([sp_of_genome.get(g, -1) for g in sorted(ids)], dtype=np.int32)
    want_path = tmp_path / "oracle_inbreed.csv"
    cols = oa.population_inbreeding(ref.filter_snp_pass(), dip, sp_of, algorithm, 0, 60000, 40, 150, 0.02, 0.9, seed=START_SEED,
                                    ped_file=(want_path, "DriverParameters", ped_rows))
    assert len(cols) >= 3
    got_text, want_text = (tmp_path / "inbreed.csv").read_text(), want_path.read_text()
    got_lines, want_lines = got_text.split("\n"), want_text.split("\n")
    assert got_lines[:2] == want_lines[:2]                                       # parameter line and column header, verbatim
    assert got_lines[1].startswith("Sample,Population,Description,SuperPopulation,Description,Relationship,Sex,Mother,Father,")
    assert len(got_lines) == len(want_lines) == 2 + len(ped_rows) + 1           # header lines, one row per genome with a PED record, final newline
    if algorithm != "Loglikelihood":
        assert got_text == want_text
    else:
        for got_line, want_line in zip(got_lines[2:], want_lines[2:]):
            g, w = got_line.split(","), want_line.split(",")
            assert g[:9] == w[:9] and len(g) == len(w)
            assert np.allclose([float(x) for x in g[9:-1]], [float(x) for x in w[9:-1]], rtol=0, atol=2e-5)


def test_gpu_inbreed_package_synthetic_file_layout(tmp_path, kgx):
    """AnalysisType=true: InbreedingOutput::writeSynthetic's file (kga_analysis_inbreed_output.cpp:308-395) -- Sample,
    SynInbreed (decoded from the synthetic genome id), one CalcInbreed per window, trailing delimiter."""
    rec, gt = sv.multiallelic_block(8, 3000, rng_seed=35, missing_af_frac=0.0, dup_records=0)
    for a in rec.af:
        a[:, 4] = a[:, 5]
    ref_path = tmp_path / "gnomad.bin"
    rio.write_records(ref_path, rec, None, ["Reference"], oa.Population.REFERENCE, "Gnomad2_1", population_id="Gnomad")
    params = dict(AnalysisType="true", OutputFile="synthetic", Algorithm="Simple", MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                  LowerWindow=0, UpperWindow=90000, LociiCount=400, SamplingDistance=20)
    res = rio.run_driver("GPU_INBREED", tmp_path, [ref_path], **params)
    assert res.returncode == 0, res.stderr
    lines = (tmp_path / "synthetic.csv").read_text().split("\n")
    assert lines[0] == "DriverParameters,Algorithm:Simple,Min_AF:0.02,Max_AF:0.9,Spacing:20,Count:400"
    assert lines[1] == "Sample,SynInbreed,CalcInbreed"
    rows = [ln.split(",") for ln in lines[2:] if ln]
    assert len(rows) == 6 * 101 and all(r[-1] == "" for r in rows)
    for r in rows[:50]:
        encoded = r[0].split("_")[1]
        want = -int(encoded[1:]) / 1e6 if encoded.startswith("N") else int(encoded) / 1e6
        assert float(r[1]) == pytest.approx(want, abs=1e-6)
    syn = np.array([float(r[1]) for r in rows])
    calc = np.array([float(r[2]) for r in rows])
    assert np.corrcoef(syn, calc)[0, 1] > 0.9


@pytest.mark.parametrize("filter_qc,filter_fws,quality_filter", [(True, True, True), (True, False, False), (False, True, False)])
def test_gpu_allele_package_with_pf7_sample_resources(tmp_path, kgx, filter_qc, filter_fws, quality_filter):
    """The package as PfEMPAnalysis runs it (kga_analysis_PfEMP.cpp:24-26,90,105,146-163): with the Pf7 sample and FWS
    resources only the genomes that pass QC / are monoclonal take part -- a genome mask evaluated on the device, the
    population is flattened and uploaded whole -- and the heterozygosity results come out in the reference's layout with
    the location summary.  FWS counts must equal the oracle's over the filtered PopulationDB, the two HeteroHomoZygous
    files must equal the oracle's byte for byte."""
    from . import pf7_text as pt
    from . import vcf_text as vt

    G, L = 150, 1500
    ids = [f"PF{i:04d}-C" for i in range(G)]
    text = vt.write_vcf_pf(L, ids, rng_seed=29)
    vcf = tmp_path / "pf.vcf"
    vcf.write_text(text)
    sample_path, fws_path, records = pt.write_resources(tmp_path, ids, rng_seed=17)
    res = rio.run_driver("GPU_ALLELE", tmp_path, [f"pf7sample:{sample_path}", f"pf7fws:{fws_path}", f"vcf:{vcf}"], VcfFlavour="Falciparum",
                         Pf7QualityFilter="TRUE" if quality_filter else "FALSE", Pf7FilterQC="TRUE" if filter_qc else "FALSE",
                         Pf7FilterFWS="TRUE" if filter_fws else "FALSE")
    assert res.returncode == 0, res.stderr

    opop = oa.Population("pf")
    opop.add_vcf_pf(text)
    if quality_filter:
        opop = opop.filter_p7()
    kept = opop.filter_pf7_genomes(sample_path, fws_path, filter_qc, filter_fws)
    kept.genome_ids = list(ids)
    kept_ids = [ids[i] for i in kept.genome_order()]
    want_ids = sorted(g for g in ids if (not filter_qc or records[g]["qc"]) and
                      (not filter_fws or (records[g]["fws"] is not None and records[g]["fws"] >= 0.95)))
    assert kept_ids == want_ids and 10 < len(kept_ids) < G

    variant_out, genome_out, vdb = kept.fws()
    header, rows = rio.read_csv(tmp_path / "VariantFWS.csv")
    assert [r[0] for r in rows] == [vdb.hgvs(i) for i in range(vdb.n_variants)]        # variants nobody kept carries are gone
    assert np.array_equal(np.array([[int(x) for x in r[-3:]] for r in rows], dtype=np.uint64), variant_out)
    header, rows = rio.read_csv(tmp_path / "GenomeFWS.csv")
    assert [r[0] for r in rows] == kept_ids
    got = np.array([[int(r[1 + 8 * b + 5 + k]) for b in range(11) for k in range(3)] for r in rows], dtype=np.uint64)
    assert np.array_equal(got.reshape(len(rows), 11, 3), genome_out)

    want_stats, want_loc = tmp_path / "oracle_stats.csv", tmp_path / "oracle_location.csv"
    assert kept.write_pfemp_location(sample_path, fws_path, want_stats, want_loc) == 0
    assert (tmp_path / "VariantLocation.csv").read_bytes() == want_loc.read_bytes()
    assert (tmp_path / "VariantStatistics.csv").read_bytes() == want_stats.read_bytes()


@pytest.mark.parametrize("binding", [{}, {"DeviceList": "0,0,0"}], ids=["one-device", "three-shards"])
def test_gpu_allele_package_population_entry_with_pf7_resources(tmp_path, kgx, binding):
    """The same through the PopulationDB entry (Variant objects delivered by a parser, here the driver's record file):
    compound offsets, indels and genomes holding a variant more than twice, under the genome filters, sharded or not."""
    from . import pf7_text as pt

    G, L = 153, 1500
    mode, source = oa.Population.UNPHASED, "Falciparum"
    rec, gt = sv.multiallelic_block(G, L, rng_seed=22)
    ids = sv.genome_ids(G, prefix="PF")
    path = tmp_path / "pop.bin"
    rio.write_records(path, rec, gt, ids, mode, source, population_id="Pf7")
    sample_path, fws_path, records = pt.write_resources(tmp_path, ids, rng_seed=31)
    res = rio.run_driver("GPU_ALLELE", tmp_path, [f"pf7sample:{sample_path}", f"pf7fws:{fws_path}", path], **binding)
    assert res.returncode == 0, res.stderr

    opop = sv.oracle_population(rec, gt, ids, mode)
    kept = opop.filter_pf7_genomes(sample_path, fws_path, True, True)
    kept.genome_ids = list(ids)
    kept_ids = [ids[i] for i in kept.genome_order()]
    assert 10 < len(kept_ids) < G
    variant_out, genome_out, vdb = kept.fws()
    header, rows = rio.read_csv(tmp_path / "VariantFWS.csv")
    assert [r[0] for r in rows] == [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert np.array_equal(np.array([[int(x) for x in r[-3:]] for r in rows], dtype=np.uint64), variant_out)
    header, rows = rio.read_csv(tmp_path / "GenomeFWS.csv")
    assert [r[0] for r in rows] == kept_ids
    got = np.array([[int(r[1 + 8 * b + 5 + k]) for b in range(11) for k in range(3)] for r in rows], dtype=np.uint64)
    assert np.array_equal(got.reshape(len(rows), 11, 3), genome_out)
    want_stats, want_loc = tmp_path / "oracle_stats.csv", tmp_path / "oracle_location.csv"
    assert kept.write_pfemp_location(sample_path, fws_path, want_stats, want_loc) == 0
    assert (tmp_path / "VariantLocation.csv").read_bytes() == want_loc.read_bytes()
    assert (tmp_path / "VariantStatistics.csv").read_bytes() == want_stats.read_bytes()


def _wide_offset_population(G, rng_seed=61):
    """A Pf7-style unphased population on one contig in which three offsets hold 28 distinct variants each (four records of
    seven alts at the same position) among ordinary one- and two-alt offsets: what HeteroHomoZygous::updateVariantAnalysisType
    (kga_analysis_PfEMP_heterozygous.cpp:61-105) walks without any cap on the variants of an offset."""
    rng = np.random.default_rng(rng_seed)
    offsets, refs, alts = [], [], []
    tails = ["C", "G", "T", "CA", "CC", "CG", "CT", "GA", "GC", "GG", "GT", "TA", "TC", "TG", "TT", "CAA", "CAC", "CAG", "CAT", "CCA", "CCC",
             "CCG", "CCT", "CGA", "CGC", "CGG", "CGT", "CTA"]
    position = 100
    for block in range(40):
        position += int(rng.integers(5, 60))
        if block % 13 == 5:                                   # a wide offset: 4 records x 7 alts, all distinct
            for r in range(4):
                offsets.append(position); refs.append("A"); alts.append(["A" + t for t in tails[7 * r:7 * r + 7]])
        elif block % 3 == 0:
            offsets.append(position); refs.append("A"); alts.append(["C", "AT"])
        else:
            offsets.append(position); refs.append("G"); alts.append(["T"])
    n = len(refs)
    af = [np.tile(np.float32(rng.uniform(0.01, 0.6)), (len(a), 6)) for a in alts]
    rec = oa.Records("Pf3D7_02_v3", offsets, refs, alts, af=af)
    gt = np.zeros((n, G, 2), dtype=np.uint8)
    for r in range(n):
        k = len(alts[r])
        carried = rng.random((G, 2)) < (0.45 if k == 7 else 0.3)
        gt[r] = np.where(carried, rng.integers(1, k + 1, (G, 2)), 0)
    return rec, gt


def test_offsets_holding_more_than_15_distinct_variants(tmp_path, kgx):
    """No cap on the variants of an offset (the 4-bit field sums of k_compound_offsets / k_offset_filters used to fail the call
    above 15): the package's VariantStatistics.csv counters against the oracle's updateVariantAnalysisType, and the four
    offset filters through the C ABI against the oracle's OffsetDB filters, on offsets holding 28 distinct variants."""
    G = 97
    rec, gt = _wide_offset_population(G)
    ids = sv.genome_ids(G, prefix="PF")
    path = tmp_path / "pop.bin"
    rio.write_records(path, rec, gt, ids, oa.Population.UNPHASED, "Falciparum", population_id="Pf7")
    res = rio.run_driver("GPU_ALLELE", tmp_path, [path])
    assert res.returncode == 0, res.stderr
    opop = sv.oracle_population(rec, gt, ids, oa.Population.UNPHASED)
    header, rows = rio.read_csv(tmp_path / "VariantStatistics.csv")
    want = opop.hethom(rec.contig)     # total,snp,indel,hom_minor,het_minor,het_ref_minor,hom_ref
    got = np.array([[int(r[2]), int(r[3]), int(r[4]), int(r[7]), int(r[8]), int(r[6]), int(r[5])] for r in rows], dtype=np.uint64)
    assert [r[0] for r in rows] == sorted(ids)
    assert np.array_equal(got, want)
    assert want[:, 3].sum() > 0 and want[:, 4].sum() > 0      # compound offsets were exercised

    # the same population through the C ABI: rows in HGVS order from the oracle's dosage matrix, groups by offset
    vdb = oa.VariantDB(opop)
    dosage = vdb.dosage()                                      # [G][V], oracle genome order
    hgvs = [vdb.hgvs(i) for i in range(vdb.n_variants)]
    import re

    offset_of = [int(re.match(r".*:g\.(\d+)", h).group(1)) for h in hgvs]
    by_offset = {}
    for row, offset in enumerate(offset_of):
        by_offset.setdefault(offset, []).append(row)
    assert max(len(v) for v in by_offset.values()) == 28
    single_bin = np.full(len(hgvs), 0xFF, dtype=np.uint8)
    members, first, count = [], [], []
    for offset, rows_of in by_offset.items():
        if len(rows_of) == 1:
            single_bin[rows_of[0]] = 0
        else:
            first.append(len(members)); count.append(len(rows_of)); members.extend(rows_of)
    pop = kgx.Population(G, len(hgvs))
    pop.load_dosage_u8(dosage)
    got_f = pop.offset_filter_counts(single_bin, members, first, count, [0] * len(first), 1)
    assert np.array_equal(got_f[:, 0, :], opop.offset_filter_counts(rec.contig))
    # and the compound-offset counters on their own: adjacent rows (HGVS order keeps an offset's rows together) and as lists
    adjacent = all(rows_of == list(range(rows_of[0], rows_of[0] + len(rows_of))) for rows_of in by_offset.values())
    listed = pop.compound_offsets_listed(members, first, count, [0] * len(first), 1)
    if adjacent:
        starts = [members[f] for f in first]
        assert np.array_equal(pop.compound_offsets(starts, count, [0] * len(first), 1), listed)
    single = pop.count_by_genome((single_bin == 0).astype(np.uint8))       # offsets of one variant: exactly one copy = het ref minor
    assert np.array_equal(listed[:, 0, 0] + single[:, 1], want[:, 5])     # het_ref_minor
    assert np.array_equal(listed[:, 0, 1] + single[:, 2] + single[:, 3], want[:, 3])   # hom_minor: distinct variants where >= 2 copies
    pop.close()


def test_gpu_inbreed_package_entropy_modes(tmp_path, kgx):
    """The three entropy modes of GPU_INBREED's iterative estimators: no StartSeed = fresh std::random_device entropy per
    window, the reference's behaviour (two runs differ in HallME's coefficients, not in the class counts); StartSeed =
    repeatable; StartPoints=Midpoint = deterministic without draws."""
    G, L = 61, 1500
    rec, gt = sv.multiallelic_block(G, L, rng_seed=47, missing_af_frac=0.02, dup_records=10)
    for a in rec.af:
        a[:, 4] = a[:, 5]
    ids = sv.genome_ids(G, prefix="NA")
    ped = [(g, "EUR") for g in ids]
    ref_path, dip_path = tmp_path / "gnomad.bin", tmp_path / "diploid.bin"
    rio.write_records(ref_path, rec, None, ["Reference"], oa.Population.REFERENCE, "Gnomad2_1", population_id="Gnomad")
    rio.write_records(dip_path, rec, gt, ids, oa.Population.PHASED, "Genome1000", population_id="Diploid", ped=ped)
    params = dict(AnalysisType="false", OutputFile="inbreed", Algorithm="HallME", MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                  LowerWindow=0, UpperWindow=60000, LociiCount=150, SamplingDistance=40)

    def run(**extra):
        res = rio.run_driver("GPU_INBREED", tmp_path, [ref_path, dip_path], **params, **extra)
        assert res.returncode == 0, res.stderr
        header, rows = rio.read_csv(tmp_path / "inbreed_detail.csv")
        counts = [[int(r[k]) for k in (2, 4, 6, 8, 10)] for r in rows]
        return counts, np.array([float(r[11]) for r in rows])

    counts_a, f_a = run()
    counts_b, f_b = run()
    assert counts_a == counts_b and np.all(np.isfinite(f_a)) and np.all(np.isfinite(f_b))
    assert not np.array_equal(f_a, f_b)                               # fresh entropy: HallME has not converged, the start shows
    assert np.abs(f_a - f_b).max() < 0.2
    _, f_c = run(StartSeed=5)
    _, f_d = run(StartSeed=5)
    assert np.array_equal(f_c, f_d)
    _, f_e = run(StartPoints="Midpoint")
    _, f_f = run(StartPoints="Midpoint")
    assert np.array_equal(f_e, f_f) and not np.array_equal(f_e, f_c)
