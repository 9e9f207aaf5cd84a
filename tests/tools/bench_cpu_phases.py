"""The phases of the path timed separately, oracle ("port" of the reference's CPU path:
nested std::map store, HGVS string keys, uint8 dosage rows, one pool task per genome) beside the GPU sweeps on the same
population, with parity checked in the same run.  Run on the GPU box:  python tests/tools/bench_cpu_phases.py [--md]

C1: 100 genomes x 50 k biallelic SNPs, one contig (the reference's own CPU-runnable case), in full.
C2 slice: 1 000 genomes x 20 k of C2's 1 M SNPs through the same sparse store (the full 1 M needs ~4e8 Variant pointers);
C2 in full through the dense tier is bench.py / tests/test_parity_gpu.py.
C5 slice: 500 genomes x 20 k multi-allelic loci: generateFrequencies + Simple over every window (the INBREED package's loop)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kgl_gene_amd import capi                                    # noqa: E402
from tests import inbreed_inputs as ii, oracle_api as oa, synth_vcf as sv   # noqa: E402

capi.init(0)
threads = oa.lib().kgo_default_threads()
print(f"host cpus {os.cpu_count()}, oracle pool threads {threads} (hardware_concurrency() - 1, the reference's default)", flush=True)
rows_md = []


def gpu_ms(fn, repeats=20):
    fn()
    times = []
    for _ in range(repeats):
        t = time.perf_counter()
        fn()
        times.append((time.perf_counter() - t) * 1e3)
    return float(np.median(times))


def allele_phases(tag, G, V):
    rec, gt, codes, af = sv.biallelic_block(G, V)
    ids = sv.genome_ids(G)
    t = time.perf_counter()
    opop = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    t_store = time.perf_counter() - t
    vdb = oa.VariantDB(opop)
    by_variant = vdb.summary_by_variant()
    by_genome = vdb.summary_by_genome()
    t = time.perf_counter()
    variant_out, genome_out, _ = opop.fws()
    t_fws = time.perf_counter() - t
    cells = G * V
    # GPU, same genotypes
    pop = capi.Population(G, V)
    pop.load_dosage2(capi.pack_dosage2(codes))
    pop.set_af(af)
    k2 = pop.allele_count_by_locus()
    rows = sv.variant_rows_in_reference_order(vdb, rec)
    carried = (k2[:, 1] + k2[:, 2] + k2[:, 3]) > 0
    order = opop.genome_order()
    ok = np.array_equal(k2[rows, :3].astype(np.uint64), by_variant)
    ok &= np.array_equal(pop.count_by_genome(carried.astype(np.uint8))[order, :3], by_genome)
    from kgl_gene_amd.fws import fws_bin_of_variant
    bins = fws_bin_of_variant(pop.get_af(), carried)
    ok &= np.array_equal(pop.count_by_genome_binned(bins, 11)[order][:, :, :3], genome_out)
    ms_k2 = gpu_ms(pop.allele_count_by_locus)
    ms_k3 = gpu_ms(lambda: pop.count_by_genome(carried.astype(np.uint8)))
    ms_bins = gpu_ms(lambda: pop.count_by_genome_binned(bins, 11))
    pop.close()
    line = (f"{tag}: {G} x {V} = {cells:.1e} cells | store {t_store:.2f} s | createVariantDB {vdb.build_seconds:.3f} s ({threads} thr) | "
            f"summaryByVariant {vdb.by_variant_seconds:.3f} s (1 thr) = {cells / vdb.by_variant_seconds:.2e} cells/s | "
            f"summaryByGenome {vdb.by_genome_seconds:.3f} s | CalcFWS (12 x createVariantDB + filters) {t_fws:.2f} s | "
            f"GPU incl. launch + D2H: K2 {ms_k2:.3f} ms, K3 {ms_k3:.3f} ms, 11 bins {ms_bins:.3f} ms | parity {'bit-exact' if ok else 'MISMATCH'}")
    print(line, flush=True)
    rows_md.append(f"| {tag} {G} × {V} | {vdb.build_seconds:.3f} s ({threads}) | {vdb.by_variant_seconds:.3f} s (1) = {cells / vdb.by_variant_seconds:.1e} cells/s | "
                   f"{vdb.by_genome_seconds:.3f} s (1) | {t_fws:.2f} s | K2 {ms_k2:.2f} ms, K3 {ms_k3:.2f} ms, 11 bins {ms_bins:.2f} ms | {'bit-exact' if ok else 'MISMATCH'} |")
    assert ok


def inbreed_phase(tag, G, L):
    rec, gt = sv.multiallelic_block(G, L, rng_seed=5, missing_af_frac=0.02, dup_records=0)
    ids = sv.genome_ids(G)
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    ref = ref.filter_snp_pass()
    dip = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    loci = ii.ReferenceLoci(rec)
    amax = max(len(a) for a in loci.alts)
    lower, upper, spacing, min_af, max_af = 0, int(loci.offsets[-1]) + 1, 1, 0.0, 1.0
    sp = np.full(G, oa.ALL, dtype=np.int32)
    counts, freqs, present, seconds = oa.inbreed_window(ref, dip, sp, "Simple", lower, upper, spacing, len(loci.offsets), min_af, max_af)
    table = loci.af_table(oa.ALL, amax)
    sel = loci.sample(table, lower, upper, spacing, min_af, max_af)
    m = capi.GenotypeMatrix(G, len(loci.offsets))
    m.load_rows(ii.encode_gt8(rec, gt, loci))
    order = dip.genome_order()
    got = m.inbreed(table[sel], "Simple", phased=True, locus_index=sel)[order]
    ok = all(np.array_equal(got[name], counts[:, k]) for k, name in
             enumerate(["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]))
    ok &= np.allclose(got["inbred_allele_sum"], freqs[:, 4], rtol=1e-10, atol=1e-12)
    ms = gpu_ms(lambda: m.inbreed(table[sel], "Simple", phased=True, locus_index=sel), repeats=10)
    m.close()
    cells = G * len(sel)
    print(f"{tag}: {G} genomes x {len(sel)} sampled loci = {cells:.1e} cells | generateFrequencies + Simple {seconds:.2f} s ({threads} thr) = {cells / seconds:.2e} cells/s | "
          f"GPU incl. table upload + D2H {ms:.2f} ms | parity {'counts bit-exact, F 1e-10' if ok else 'MISMATCH'}", flush=True)
    rows_md.append(f"| {tag} {G} × {len(sel)} loci | — | — | — | generateFrequencies + Simple: {seconds:.2f} s ({threads}) = {cells / seconds:.1e} cells/s | K5 {ms:.2f} ms | {'counts bit-exact, F 1e-10' if ok else 'MISMATCH'} |")
    assert ok


allele_phases("C1", 100, 50_000)
allele_phases("C2 slice", 1000, 20_000)
inbreed_phase("C5 slice", 500, 20_000)
if "--md" in sys.argv:
    print("\n| config | createVariantDB (threads) | summaryByVariant | summaryByGenome | CalcFWS / inbreeding | GPU (wall per call) | parity |\n|---|---|---|---|---|---|---|")
    print("\n".join(rows_md))
