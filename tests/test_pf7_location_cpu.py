"""The Pf7 sample resources and the location analysis of HeteroHomoZygous on the host: GpuHeteroHomoZygous (fed the oracle's
own per-genome counters, no device) must write the reference's two files byte for byte as the oracle's restatement does."""
import numpy as np
import pytest

from . import host_api as ha
from . import oracle_api as oa
from . import pf7_text as pt
from . import vcf_text as vt

CONTIGS = ["Pf3D7_01_v3", "Pf3D7_02_v3", "Pf3D7_MIT_v3", "Pf3D7_API_v3"]


def _oracle_counters(pop):
    genomes = sorted(pop.genome_ids)
    order = pop.genome_order()
    names = [pop.genome_ids[i] for i in order]
    records = []
    for contig in CONTIGS:
        want, present = pop.hethom(contig), pop.hethom_present(contig)
        for g, name in enumerate(names):
            if present[g]:
                c = want[g]           # total, snp, indel, hom_minor, het_minor, het_ref_minor, hom_ref
                records.append((name, contig, tuple(int(x) for x in c)))
    return records


@pytest.mark.parametrize("radius_km", [0.0, 150.0, 20000.0])
@pytest.mark.parametrize("filter_qc,filter_fws", [(True, True), (False, False), (True, False)])
def test_location_files_match_oracle(tmp_path, radius_km, filter_qc, filter_fws):
    G = 90
    ids = [f"PF{i:04d}-C" for i in range(G)]
    text = vt.write_vcf_pf(400, ids, rng_seed=33)
    sample_path, fws_path, _ = pt.write_resources(tmp_path, ids)
    opop = oa.Population("pf")
    opop.add_vcf_pf(text)
    kept = opop.filter_pf7_genomes(sample_path, fws_path, filter_qc, filter_fws)
    kept.genome_ids = [g for g in ids]     # genome_order indexes the caller's list; only the kept ones appear in it
    want_stats, want_loc = tmp_path / "o_stats.csv", tmp_path / "o_loc.csv"
    assert kept.write_pfemp_location(sample_path, fws_path, want_stats, want_loc, radius_km) == 0

    records = _oracle_counters(kept)
    got_stats, got_loc = tmp_path / "g_stats.csv", tmp_path / "g_loc.csv"
    assert ha.pfemp_location_write(sample_path, fws_path, records, got_stats, got_loc, radius_km) == 0
    assert got_loc.read_bytes() == want_loc.read_bytes()
    assert got_stats.read_bytes() == want_stats.read_bytes()
    # the files say something: sites and countries, a site above the 20-sample bar, non-zero F_IS somewhere
    loc = [line.split(",") for line in want_loc.read_text().splitlines()[1:]]
    assert {row[1] for row in loc} == {"City", "Country"}
    assert any(int(row[7]) >= 20 for row in loc if row[1] == "City")
    stats = [line.split(",") for line in want_stats.read_text().splitlines()[1:]]
    assert any(float(row[2]) != 0.0 for row in stats)
    if filter_qc and filter_fws:
        assert 0 < len(stats) < G


def test_genome_filter_keeps_the_documented_genomes(tmp_path):
    """QC pass and FWS >= 0.95, samples without a record or a value dropped: the oracle's filtered population holds exactly
    the genomes the resource files say."""
    G = 60
    ids = [f"PF{i:04d}-C" for i in range(G)]
    text = vt.write_vcf_pf(120, ids, rng_seed=3)
    sample_path, fws_path, records = pt.write_resources(tmp_path, ids, rng_seed=9)
    opop = oa.Population("pf")
    opop.add_vcf_pf(text)
    kept = opop.filter_pf7_genomes(sample_path, fws_path, True, True)
    kept.genome_ids = list(ids)
    names = [ids[i] for i in kept.genome_order()]
    want = sorted(g for g in ids if records[g]["qc"] and records[g]["fws"] is not None and records[g]["fws"] >= 0.95)
    assert names == want and 0 < len(want) < G


def test_bad_resource_files_are_refused(tmp_path):
    short = tmp_path / "short.txt"
    short.write_text("Sample\tStudy\nA\tB\n")                       # 2 columns where 17 are required
    fws = tmp_path / "fws.txt"
    fws.write_text("Sample\tFws\nA\t0.99\n")
    assert ha.pfemp_location_write(short, fws, [], tmp_path / "a.csv", tmp_path / "b.csv") == -1
    assert ha.pfemp_location_write(tmp_path / "missing.txt", fws, [], tmp_path / "a.csv", tmp_path / "b.csv") == -1
