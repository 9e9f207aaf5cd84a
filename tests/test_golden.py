"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py with the oracle).
CPU: the oracle still reproduces them (dense tier on the stored inputs).  GPU: the HIP path matches them."""
from pathlib import Path

import numpy as np
import pytest

from kgl_gene_amd import capi
from kgl_gene_amd.fws import fws_bin_of_variant

from . import oracle_api as oa

GOLD = Path(__file__).resolve().parent / "golden"


def test_oracle_dense_tier_reproduces_allele_golden():
    g = np.load(GOLD / "allele_48x600.npz")
    G = int(g["n_genomes"])
    codes = capi.unpack_dosage2(g["packed"], G)                 # [V][G] in caller order
    rows, order = g["reference_row_order"], g["genome_order"]
    dense = oa.Dense(np.ascontiguousarray(codes[rows][:, order].T))
    assert np.array_equal(dense.summary_by_variant(), g["summary_by_variant"])
    assert np.array_equal(dense.summary_by_genome(), g["summary_by_genome"])
    assert g["summary_by_variant"].sum(1).tolist() == [G] * len(rows)           # conservation identity
    assert np.array_equal(g["summary_by_variant"].sum(0), g["population_summary"])


@pytest.mark.gpu
def test_gpu_matches_allele_golden(kgx):
    g = np.load(GOLD / "allele_48x600.npz")
    G, V = int(g["n_genomes"]), g["packed"].shape[0]
    pop = kgx.Population(G, V)
    pop.load_dosage2(g["packed"])
    pop.set_af(g["af"])
    k2 = pop.allele_count_by_locus()
    rows, order = g["reference_row_order"], g["genome_order"]
    assert np.array_equal(k2[rows, :3].astype(np.uint64), g["summary_by_variant"])
    carried = (k2[:, 1] + k2[:, 2] + k2[:, 3]) > 0
    assert np.array_equal(pop.count_by_genome(carried.astype(np.uint8))[order, :3], g["summary_by_genome"])
    bins = fws_bin_of_variant(g["af"], carried)
    assert np.array_equal(pop.count_by_genome_binned(bins, 11)[order][:, :, :3], g["fws_genome_bins"])
    pop.close()


def test_reference_start_points_match_golden():
    """CPU (host code only): the start points the product draws for the reference's restarts -- the fifth draw of every
    genome's std::mt19937_64 stream -- are the ones the oracle drew when the golden file was made, and the oracle still
    draws them."""
    g = np.load(GOLD / "inbreed_40x700.npz")
    seed, G = int(g["start_seed"]), g["gt8"].shape[1]
    for algorithm in ("HallME", "Loglikelihood"):
        assert np.array_equal(capi.reference_starts(algorithm, seed, G), g[f"start_{algorithm}"])
        assert np.array_equal(oa.restart_draws(algorithm, seed, G)[:, 4], g[f"start_{algorithm}"])
        # streams are per genome: a range of genomes draws the same points as the whole
        assert np.array_equal(capi.reference_starts(algorithm, seed, 7, first_stream=11), g[f"start_{algorithm}"][11:18])
    lo, hi = g["start_HallME"].min(), g["start_HallME"].max()
    assert 0.0 < lo and hi <= 0.5 and -0.5 < g["start_Loglikelihood"].min() and g["start_Loglikelihood"].max() <= 0.5
    # seed 0 = std::random_device, the reference's production entropy: two calls differ
    assert not np.array_equal(capi.reference_starts("HallME", 0, 8), capi.reference_starts("HallME", 0, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm,tol", [("Simple", 1e-10), ("RitlandLocus", 1e-10), ("HallME", 1e-9), ("Loglikelihood", 2e-6)])
def test_gpu_matches_inbreed_golden(kgx, algorithm, tol):
    g = np.load(GOLD / "inbreed_40x700.npz")
    gt8 = g["gt8"]
    m = kgx.GenotypeMatrix(gt8.shape[1], gt8.shape[0])
    m.load_rows(gt8)
    sel = g["selected"]
    start = None
    if f"start_{algorithm}" in g:                              # stored in genome-id order, the order of the oracle's fan-out
        start = np.empty(gt8.shape[1])
        start[g["genome_order"]] = g[f"start_{algorithm}"]
    got = m.inbreed(g["af_table"][sel], algorithm, phased=True, locus_index=sel, start=start)[g["genome_order"]]
    counts, freqs = g[f"counts_{algorithm}"], g[f"freqs_{algorithm}"]
    for k, name in enumerate(["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]):
        assert np.array_equal(got[name], counts[:, k])
    for k, name in enumerate(["major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"]):
        assert np.allclose(got[name], freqs[:, k], rtol=1e-12, atol=1e-12)
    assert np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max() <= tol
    m.close()


def test_pf_vcf_flattener_matches_golden():
    """CPU: the product's Pf flattener (host C++) against the committed oracle outputs for a committed VCF text."""
    from . import host_api as ha

    g = np.load(GOLD / "vcf_cases.npz")
    text = str(g["pf_text"][0])
    for tag, quality_filter in (("raw", False), ("p7", True)):
        flat = ha.FlatVcf(text, 3, flavour="Falciparum", quality_filter=quality_filter)
        assert flat.hgvs == g[f"pf_{tag}_hgvs"].tolist()
        assert flat.genome_ids == g[f"pf_{tag}_genomes"].tolist()
        assert np.array_equal(capi.unpack_dosage2(flat.packed, flat.G), np.minimum(g[f"pf_{tag}_dosage"].T, 3))
    assert len(g["pf_p7_hgvs"]) < len(g["pf_raw_hgvs"])
    # and the oracle still reproduces its own committed parse
    o = oa.Population("pf")
    o.add_vcf_pf(text)
    vdb = oa.VariantDB(o)
    assert [vdb.hgvs(i) for i in range(vdb.n_variants)] == g["pf_raw_hgvs"].tolist()
    assert np.array_equal(vdb.dosage(), g["pf_raw_dosage"])


def test_variant_sort_indexes_match_golden():
    """CPU: the rsid / Ensembl indexes on columns against the committed oracle outputs for a committed VCF text; and the
    oracle still reproduces them."""
    from . import host_api as ha
    from . import test_variant_sort_cpu as ts

    g = np.load(GOLD / "variant_sort.npz")
    for flavour in ("MonoGenome", "Genome1000"):
        text = str(g[f"{flavour}_text"][0])
        population = ts.oracle_population(text, flavour)
        for what in ts.KINDS_ALL:
            want = [tuple(line.split("\t")) for line in g[f"{flavour}_{what}"].tolist()]
            assert ha.variant_sort(text, flavour, what, threads=2) == want, (flavour, what)
            assert population.variant_sort(what) == want, (flavour, what)
            assert what == "non_ensembl" or len(want) > 10
        listed = g[f"{flavour}_filter_list"].tolist()
        want = [tuple(line.split("\t")) for line in g[f"{flavour}_filter"].tolist()]
        assert ha.variant_sort(text, flavour, "filter", listed) == want and len(want) > 0


@pytest.mark.gpu
def test_gpu_inbreed_from_vcf_matches_golden(kgx, tmp_path):
    """GPU: GPU_INBREED fed the two committed VCF texts reproduces the committed oracle window (Simple)."""
    from . import records_io as rio

    g = np.load(GOLD / "vcf_cases.npz")
    (tmp_path / "gnomad.vcf").write_text(str(g["ref_text"][0]))
    (tmp_path / "kg.vcf").write_text(str(g["kg_text"][0]))
    genomes = g["kg_genomes"].tolist()
    (tmp_path / "ped.txt").write_text("".join(f"{name}\tALL\n" for name in genomes))
    res = rio.run_driver("GPU_INBREED", tmp_path, [f"vcf:Gnomad2_1:{tmp_path / 'gnomad.vcf'}", f"vcf:Genome1000:{tmp_path / 'kg.vcf'}",
                                                    f"ped:{tmp_path / 'ped.txt'}"],
                         AnalysisType="false", OutputFile="inbreed", Algorithm="Simple", MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                         LowerWindow=0, UpperWindow=10**9, LociiCount=100, SamplingDistance=10)
    assert res.returncode == 0, res.stderr
    header, rows = rio.read_csv(tmp_path / "inbreed_detail.csv")
    got = {(r[0], r[1]): ([int(r[2]), int(r[4]), int(r[6]), int(r[8]), int(r[10])], [float(r[3]), float(r[5]), float(r[7]), float(r[9]), float(r[11])])
           for r in rows}
    n = 0
    for c, ident in enumerate(g["kg_column_ident"].tolist()):
        for k, name in enumerate(genomes):
            if not g["kg_present"][c, k]:
                assert (ident, name) not in got
                continue
            counts, freqs = got[(ident, name)]
            assert counts == g["kg_counts"][c, k].tolist(), (ident, name)
            assert np.allclose(freqs[:4], g["kg_freqs"][c, k, :4], rtol=1e-12, atol=1e-12)
            assert abs(freqs[4] - g["kg_freqs"][c, k, 4]) <= 1e-10
            n += 1
    assert n == len(got) and n > 0
