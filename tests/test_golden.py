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


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm,tol", [("Simple", 1e-10), ("RitlandLocus", 1e-10), ("HallME", 1e-9), ("Loglikelihood", 1e-5)])
def test_gpu_matches_inbreed_golden(kgx, algorithm, tol):
    g = np.load(GOLD / "inbreed_40x700.npz")
    gt8 = g["gt8"]
    m = kgx.GenotypeMatrix(gt8.shape[1], gt8.shape[0])
    m.load_rows(gt8)
    sel = g["selected"]
    got = m.inbreed(g["af_table"][sel], algorithm, phased=True, locus_index=sel)[g["genome_order"]]
    counts, freqs = g[f"counts_{algorithm}"], g[f"freqs_{algorithm}"]
    for k, name in enumerate(["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]):
        assert np.array_equal(got[name], counts[:, k])
    for k, name in enumerate(["major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"]):
        assert np.allclose(got[name], freqs[:, k], rtol=1e-12, atol=1e-12)
    assert np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max() <= tol
    m.close()
