"""Regenerate the committed golden vectors with the oracle (CPU):  python -m tests.golden.make_golden

The reference holds no fixtures for this path, so these are ORACLE outputs (parity unpinned, see oracle/kgo_core.h):
they freeze the oracle's behaviour and give the GPU path a reference-free target.  Inputs are stored next to the
expected outputs so the files are self-contained data (no reference source, no generator needed to read them)."""
from pathlib import Path

import numpy as np

from kgl_gene_amd import capi
from tests import inbreed_inputs as ii
from tests import oracle_api as oa
from tests import synth_vcf as sv

HERE = Path(__file__).resolve().parent


def allele_case():
    G, V = 48, 600
    rec, gt, codes, af = sv.biallelic_block(G, V)
    af = af.copy()
    af[:6] = [0.05, 0.5, 1.0, np.nan, 0.45, 0.0]
    rec.af = [np.tile(np.float32(a), (1, 6)) for a in af]
    ids = sv.genome_ids(G)
    pop = sv.oracle_population(rec, gt, ids, oa.Population.UNPHASED)
    variant_out, genome_out, vdb = pop.fws()
    rows = sv.variant_rows_in_reference_order(vdb, rec)
    np.savez_compressed(HERE / "allele_48x600.npz", n_genomes=G, packed=capi.pack_dosage2(codes), af=af,
                        reference_row_order=rows, genome_order=pop.genome_order(),
                        summary_by_variant=vdb.summary_by_variant(), summary_by_genome=vdb.summary_by_genome(),
                        population_summary=vdb.population_summary(), fws_genome_bins=genome_out,
                        hgvs_first=np.array([vdb.hgvs(0)]), hethom=pop.hethom(rec.contig))


def inbreed_case():
    G, L = 40, 700
    rec, gt = sv.multiallelic_block(G, L, rng_seed=3, missing_af_frac=0.03, dup_records=0)   # duplicate records can put two same-phase copies of one variant in a genome, which the gt8 encoding rejects
    ids = sv.genome_ids(G)
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    ref_f = ref.filter_snp_pass()
    dip = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    loci = ii.ReferenceLoci(rec)
    amax = max(len(a) for a in loci.alts)
    table = loci.af_table(oa.ALL, amax)
    args = dict(lower=100, upper=30_000, spacing=30, min_af=0.02, max_af=0.9)
    sel = loci.sample(table, args["lower"], args["upper"], args["spacing"], args["min_af"], args["max_af"])
    out = dict(gt8=ii.encode_gt8(rec, gt, loci, phased_order=True), af_table=table, selected=sel, genome_order=dip.genome_order(),
               offsets=loci.offsets)
    for algo in ("Simple", "RitlandLocus", "HallME", "Loglikelihood"):
        counts, freqs, present, _ = oa.inbreed_window(ref_f, dip, np.full(G, oa.ALL, dtype=np.int32), algo, args["lower"], args["upper"],
                                                      args["spacing"], 1000, args["min_af"], args["max_af"], seed=oa.FIXED_STARTS)
        assert present.all()
        out[f"counts_{algo}"] = counts
        out[f"freqs_{algo}"] = freqs
    np.savez_compressed(HERE / "inbreed_40x700.npz", **out)


if __name__ == "__main__":
    allele_case()
    inbreed_case()
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")
