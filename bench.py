#!/usr/bin/env python3
"""Headline benchmark: the allele-frequency sweep (K2 = summaryByVariant for every variant, + the
all-reduce of per-variant counts across genome shards, + the AF epilogue) over a synthetic population
resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over this rank's genome shard.  Workload (config.workload):
  c3  (default) 10,000 genomes x 10,000,000 biallelic SNPs PER GPU  (BASELINE.json configs[2]; weak scaling)
  c4            100,000 genomes x 10M SNPs split over the N ranks    (configs[3]; needs N >= 2)
  c2            1,000 genomes x 1M SNPs per GPU                      (configs[1]; fits the 256 MiB L3)
  c5            10,000 genomes x 5,000,000 multi-allelic loci per GPU: the inbreeding sweep (K5 + the estimator named by
                --algorithm; configs[4]); genomes are independent, so N ranks are N shards with no exchange at all
One JSON line is printed by rank 0.  At N = 1 the default (c3) run also carries, under "aux", the two other sweeps of
the path on the same box in the same run -- the 11-bin by-genome sweep (K3) on the C3 population and the inbreeding
sweep + Simple (K5) on the C5 population -- each with its own roofline and its own CPU baseline (--no-aux skips them).
PyTorch is plumbing only (device tensors, RCCL all-reduce).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling

WORKLOADS = {
    "c2": dict(genomes_per_gpu=1_000, variants=1_000_000, label="C2: 1k genomes x 1M biallelic SNPs per GPU"),
    "c3": dict(genomes_per_gpu=10_000, variants=10_000_000, label="C3: 10k genomes x 10M biallelic SNPs per GPU"),
    "c4": dict(total_genomes=100_000, variants=10_000_000, label="C4: 100k genomes x 10M biallelic SNPs sharded over the ranks"),
    "c5": dict(genomes_per_gpu=10_000, variants=5_000_000, label="C5: 10k genomes x 5M multi-allelic loci per GPU, inbreeding sweep"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--genomes", type=int, default=0, help="override genomes per GPU")
    ap.add_argument("--variants", type=int, default=0, help="override variant rows")
    ap.add_argument("--seed", type=int, default=1111)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the K3 / C5 side measurements of the default run")
    ap.add_argument("--cpu-sample-variants", type=int, default=300_000)      # x 10k genomes = 3e9 cells: ~12 s of the port
    ap.add_argument("--aux-genomes", type=int, default=10_000, help="genomes of the aux C5 population")
    ap.add_argument("--aux-loci", type=int, default=5_000_000, help="loci of the aux C5 population")
    ap.add_argument("--c4-genomes", type=int, default=0,
                    help="total genomes of the strong-scaled aux job of an N > 1 run (default: C4's 100,000 when the headline shape is C3's)")
    ap.add_argument("--algorithm", choices=["Simple", "RitlandLocus", "HallME", "Loglikelihood"], default="Simple",
                    help="estimator of the c5 workload")
    return ap.parse_args()


def workload_label(args, wl, genomes, variants) -> str:
    """The preset's label only when the preset's shape was run; an overridden shape names itself."""
    preset_g = wl.get("genomes_per_gpu")
    if (not args.genomes or args.genomes == preset_g) and (not args.variants or args.variants == wl["variants"]):
        return wl["label"]
    what = "multi-allelic loci, inbreeding sweep" if args.workload == "c5" else "biallelic SNPs"
    return f"custom ({args.workload} shape overridden): {genomes} genomes x {variants} {what} per GPU"


def k2_traffic(G, V):
    """HBM bytes per K2 launch from the committed rocprofv3 --pmc passes (profiles/): a constant of an earlier profiling
    run of this exact shape, not something a bench run can measure -- hence traffic_source beside it."""
    tf = ROOT / "profiles" / "k2_traffic.json"
    if tf.exists():
        try:
            rec = json.loads(tf.read_text()).get(f"{G}x{V}")
            if rec:
                return rec["hbm_bytes_per_launch"], (f"profiles/k2_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc "
                                                     f"passes over this shape ({rec.get('profile', 'earlier profiling run')}); not measured in this run")
        except Exception:
            pass
    return None, "no rocprofv3 --pmc pass committed for this shape"


def committed_traffic(label, workload):
    """HBM bytes per launch of an aux sweep from profiles/traffic.json (scripts/summarize_round.py: the FETCH_SIZE and
    WRITE_SIZE passes of the round's profiling run over this same workload), or None."""
    tf = ROOT / "profiles" / "traffic.json"
    try:
        rec = json.loads(tf.read_text()).get(f"{label}:{workload}") if tf.exists() else None
    except Exception:
        rec = None
    if rec and "hbm_bytes_per_launch" in rec:
        return rec["hbm_bytes_per_launch"], (f"profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over this "
                                             f"workload ({rec.get('profile', 'earlier profiling run')}, kernel {rec.get('kernel')}); not measured in this run")
    return None, "no rocprofv3 --pmc pass committed for this workload"


def cpu_baseline(capi, pop, G, V, k2_host_sample_rows, sample_variants, seed):
    """The oracle's dense tier (the reference's summaryByVariant column walk over VariantDBGenomeData,
    single-threaded as in CalcFWS::updateVariantFWSMap) timed on this box's host cores, on the first
    `sample_variants` rows x all genomes of the same population.  Also a parity check of that block.
    Returns (record, dense oracle object, sample width) -- the by-genome aux leg reuses the matrix."""
    from tests import oracle_api as oa

    nv = min(sample_variants, V)
    packed = pop.read_dosage2(0, nv)
    codes = capi.unpack_dosage2(packed, G)             # [nv][G]
    dosage = np.ascontiguousarray(codes.T)             # VariantDBGenomeData layout [G][nv]
    del codes, packed
    dense = oa.Dense(dosage)
    want = dense.summary_by_variant()
    seconds = dense.seconds
    ok = bool(np.array_equal(k2_host_sample_rows[:, :3].astype(np.uint64), want))
    # second, clearly labelled figure: a tuned CPU sweep of the same 2-bit rows, every host thread
    n_fast = min(V, 10 * nv)
    fast_rows = pop.read_dosage2(0, n_fast)
    fast_counts, fast_seconds, fast_threads = None, float("inf"), 0
    for threads in sorted({16, 64, os.cpu_count() or 16}):          # the box may grant fewer cores than it shows
        counts_t, seconds_t, used_t = oa.fast_count_by_variant(fast_rows, G, threads=threads)
        if seconds_t < fast_seconds:
            fast_counts, fast_seconds, fast_threads = counts_t, seconds_t, used_t
    fast_ok = bool(np.array_equal(fast_counts[:nv], k2_host_sample_rows))
    del fast_rows
    record = {
        "value": G * nv / seconds,
        "unit": "variants·genomes/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {nv} variants x all {G} genomes of the same population ({G * nv:.3g} cells, {seconds:.1f} s); "
                  f"oracle dense tier = reference summaryByVariant loop (single-threaded in the reference); "
                  f"host has {os.cpu_count()} cpus; block parity vs GPU: {'bit-exact' if ok else 'MISMATCH'}",
        "parity_ok": ok,
        "multithreaded_path": multithreaded_reference_path(capi, seed),
        "optimised_cpu": {
            "value": G * n_fast / fast_seconds, "unit": "variants·genomes/s", "cores": fast_threads, "kind": "not the reference's algorithm",
            "sample": f"first {n_fast} variants x all {G} genomes, same 2-bit rows as the GPU, 64-bit popcounts on {fast_threads} threads, "
                      f"best of 3 ({fast_seconds * 1e3:.1f} ms); parity vs GPU: {'bit-exact' if fast_ok else 'MISMATCH'}",
        },
    }
    return record, dense, nv


def multithreaded_reference_path(capi, seed, G=1000, V=20_000):
    """north_star's "multithreaded CPU path" end to end on a slice the pointer-chasing store can hold: the sparse
    PopulationDB of G x V of the same synthetic population -> createVariantDB (processAll_MT, hw - 1 threads, one task per
    genome, HGVS string keys: kgl_variant_db_variant.cpp:11-123) -> summaryByVariant for every variant (serial, as
    CalcFWS::updateVariantFWSMap calls it).  The store itself is the parser's output and is not timed."""
    from tests import oracle_api as oa

    rows, af = capi.synth_biallelic_host(seed, 0, G, 0, V)
    codes = capi.unpack_dosage2(rows, G)                                   # [V][G]
    rng = np.random.default_rng(7)
    offsets = np.cumsum(rng.integers(1, 51, V)).astype(np.uint64)
    ref_code = rng.integers(0, 4, V).astype(np.uint8)
    alt_code = ((ref_code + rng.integers(1, 4, V)) % 4).astype(np.uint8)
    gt = np.zeros((V, G, 2), dtype=np.uint8)
    het_phase = rng.integers(0, 2, (V, G)).astype(np.uint8)
    gt[..., 0] = np.where(codes == 2, 1, np.where((codes == 1) & (het_phase == 0), 1, 0))
    gt[..., 1] = np.where(codes == 2, 1, np.where((codes == 1) & (het_phase == 1), 1, 0))
    opop = oa.Population("synthetic")
    opop.add_genomes([f"HG{i:06d}" for i in range(G)])
    opop.add_records_coded("chr1", offsets, ref_code, np.ones(V, dtype=np.uint8), alt_code, np.repeat(af[:, None], 6, axis=1), gt,
                           oa.Population.PHASED)
    vdb = oa.VariantDB(opop)                                               # createVariantDB, timed inside the oracle
    by_variant = vdb.summary_by_variant()
    threads = int(oa.lib().kgo_pool_threads(G))
    # parity of the slice against the GPU sweep of the same genotypes
    small = capi.Population(G, V)
    small.load_dosage2(rows)
    k2 = small.allele_count_by_locus()
    small.close()
    rec_idx, _ = vdb.variant_keys()
    ok = bool(np.array_equal(k2[rec_idx.astype(np.int64), :3].astype(np.uint64), by_variant))
    seconds = vdb.build_seconds + vdb.by_variant_seconds
    return {
        "value": G * V / seconds, "unit": "variants·genomes/s", "cores": threads, "kind": "port",
        "phases_s": {"createVariantDB": round(vdb.build_seconds, 3), "summaryByVariant": round(vdb.by_variant_seconds, 3)},
        "sample": f"{G} genomes x {V} variants of the same population through the sparse store ({G * V:.3g} cells): createVariantDB on {threads} "
                  f"pool threads (hardware_concurrency() - 1 capped by the genome count, the reference's rule) + serial summaryByVariant; "
                  f"parity vs GPU: {'bit-exact' if ok else 'MISMATCH'}",
        "parity_ok": ok,
    }


def aux_by_genome(capi, torch, pop, counts, G, V, dense, nv):
    """K3: CalcFWS's eleven allele-frequency bins x every genome (kga_analysis_PfEMP_FWS.cpp:15-38,72-101) on the population
    the headline ran on: one kgx_count_by_genome_binned call = the reference's 11 x (viewFilter + createVariantDB +
    summaryByGenome).  Kernel time from HIP events inside the library; CPU leg: the oracle's summaryByGenome loop over each
    bin's variants of the dense sample (single-threaded, as the reference's), parity asserted on that block."""
    from kgl_gene_amd.fws import FWS_BINS, fws_bin_of_variant

    carried = ((counts[:, 1] + counts[:, 2] + counts[:, 3]) > 0).cpu().numpy()
    af = pop.get_af()
    bins = fws_bin_of_variant(af, carried)
    selected = int((bins != 0xFF).sum())
    # The bins are decided on the device from the AF column (the P7FrequencyFilter pair as a per-launch predicate); a row
    # nobody carries is not in the reference's store, so it carries no AF value here.
    af_carried = af.copy()
    af_carried[~carried] = np.nan
    pop.set_af(af_carried)
    edges = [lo for lo, _ in FWS_BINS] + [FWS_BINS[-1][1]]
    walls, kernels = [], []
    for i in range(6):
        t = time.perf_counter()
        by_bin = pop.count_by_genome_af_bins(edges)
        if i:
            walls.append((time.perf_counter() - t) * 1e3)
            kernels.append(capi.count_by_genome_last_ms())
    wall_ms, kernel_ms = float(np.median(walls)), float(np.median(kernels))
    algorithmic = selected * ((G + 3) // 4) + 32 * G * 11
    achieved = algorithmic / (kernel_ms * 1e-3) / 1e9
    k3_workload = f"K3 on the headline population: {G} genomes x {selected} binned variants of {V}"
    k3_traffic, k3_traffic_source = committed_traffic("K3", k3_workload)
    record = {
        "metric": "variants·genomes/sec (by-genome sweep, 11 FWS allele-frequency bins)",
        "value": G * selected / (wall_ms * 1e-3), "unit": "variants·genomes/s", "ms_per_call": wall_ms, "calls": len(walls),
        "config": {"workload": k3_workload,
                   "note": "one kgx_count_by_genome_af_bins call: bins evaluated on the device from the AF column, rows grouped by bin "
                           "on the device, result downloaded; equal to the host-binned call: " + str(bool(np.array_equal(by_bin, pop.count_by_genome_binned(bins, 11))))},
        "roofline": {"bound": "hbm", "kernel": "k_count_by_genome", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": k3_traffic, "traffic_source": k3_traffic_source,
                     "algorithmic_bytes_per_launch": algorithmic, "kernel_ms": kernel_ms},
        "cpu_baseline": None,
    }
    if dense is not None:
        sample_bins = bins.copy()
        sample_bins[nv:] = 0xFF
        got = pop.count_by_genome_binned(sample_bins, 11)
        seconds, ok = 0.0, True
        for b in range(11):
            want = dense.summary_by_genome((sample_bins[:nv] == b).astype(np.uint8))
            seconds += dense.seconds
            ok = ok and bool(np.array_equal(got[:, b, :3], want))
        cells = G * int((sample_bins != 0xFF).sum())
        record["cpu_baseline"] = {
            "value": cells / seconds, "unit": "variants·genomes/s", "cores": 1, "kind": "port",
            "sample": f"the 11 bins of the first {nv} variants x all {G} genomes ({cells:.3g} cells, {seconds:.1f} s): oracle dense tier = "
                      f"the reference's summaryByGenome row walk per bin (serial in the reference); parity vs GPU: {'bit-exact' if ok else 'MISMATCH'}",
            "parity_ok": ok}
    return record


def aux_genome_mask(capi, torch, pop, counts, G, V, sweep_bytes, seed):
    """K2 under a genome mask (kgx_population_set_genome_mask: the Pf7 QC / monoclonal genome lists of FilterPf7 as a
    per-launch predicate instead of a re-flattened population): same rows, one cached mask row more.  Checked in-run
    against the unmasked counts of the kept genomes' complement being absent: row sums == genomes kept."""
    keep = np.random.default_rng(seed).random(G) < 0.7
    pop.set_genome_mask(keep)
    ms = pop.allele_count_timed(counts.data_ptr(), torch.cuda.current_stream().cuda_stream, 2, 10)
    torch.cuda.synchronize()
    rows = counts[:4096].cpu().numpy().view(np.uint32)
    ok = bool(np.all(rows.astype(np.uint64).sum(1) == int(keep.sum())))
    pop.set_genome_mask(None)
    kernel_ms = float(np.median(ms))
    achieved = sweep_bytes / (kernel_ms * 1e-3) / 1e9
    return {
        "metric": "variants·genomes/sec (allele-freq sweep under a genome mask)",
        "value": int(keep.sum()) * V / (kernel_ms * 1e-3), "unit": "variants·genomes/s (genomes kept)", "ms_per_call": kernel_ms, "calls": len(ms),
        "config": {"workload": f"K2 on the headline population with {int(keep.sum())} of {G} genomes kept",
                   "check": "row sums == genomes kept on the first 4096 variants: " + ("ok" if ok else "MISMATCH")},
        "roofline": {"bound": "hbm", "kernel": "k_allele_count<..., MASKED>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": sweep_bytes + (G + 3) // 4,
                     "kernel_ms": kernel_ms},
        "cpu_baseline": None,
    }


def aux_inbreeding(args, capi, torch, dev, cpu):
    """C5 (BASELINE.json configs[4]) on this GPU: generateFrequencies + processSimple for every genome over every locus
    (kga_analysis_inbreed_freq.cpp:425-583, _calc.cpp:318-365) = one kgx_inbreed call; the AF table is resident in HBM.
    CPU leg: the oracle's processResults (one pool task per genome, hw - 1 threads) on a >= 1e8-cell slice of the same
    population, regenerated by the host twin; parity asserted on that slice in the same run."""
    from tests import oracle_api as oa
    from tests import synth_vcf as sv

    G, L = args.aux_genomes, args.aux_loci
    m = capi.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(args.seed, 0, 0)
    n_sel, amax = table.shape
    sweep_bytes = int(capi.lib().kgx_gt8_sweep_bytes(G, L, amax))
    table_dev = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(dev)
    walls, sweeps, kernels = [], [], []
    res = None
    for i in range(7):
        t = time.perf_counter()
        res = m.inbreed_resident(table_dev.data_ptr(), n_sel, amax, "Simple", phased=True)
        if i >= 2:
            walls.append((time.perf_counter() - t) * 1e3)
            sweeps.append(capi.inbreed_last_sweep_ms())
            kernels.append(capi.inbreed_last_kernel_ms())
    wall_ms, sweep_ms, kernel_ms = float(np.median(walls)), float(np.median(sweeps)), float(np.median(kernels))
    achieved = sweep_bytes / (kernel_ms * 1e-3) / 1e9
    label = WORKLOADS["c5"]["label"] if (G, L) == (10_000, 5_000_000) else f"custom: {G} genomes x {L} multi-allelic loci, inbreeding sweep"
    k5_traffic, k5_traffic_source = committed_traffic("K5", label)
    record = {
        "metric": "genomes·loci/sec (inbreeding sweep + Simple)",
        "value": G * L / (wall_ms * 1e-3), "unit": "genomes·loci/s", "ms_per_call": wall_ms, "calls": len(walls), "dtype": "u8 classes, f64 sums",
        "config": {"workload": label, "genomes": G, "loci": L, "algorithm": "Simple", "layout": "gt8 allele-index bytes, locus-major",
                   "mean_F": float(res["inbred_allele_sum"].mean())},
        "roofline": {"bound": "hbm", "kernel": "k_inbreed_eval_lut<4> (the frequency sweep's table pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": k5_traffic, "traffic_source": k5_traffic_source,
                     "algorithmic_bytes_per_launch": sweep_bytes, "kernel_ms": kernel_ms,
                     "kernel_ms_statistic": f"median of {len(kernels)} launches (HIP events on the launch stream)",
                     "sweep_ms_with_locus_helpers": sweep_ms},
        "cpu_baseline": None,
    }
    if cpu:
        Gs, Ls = min(G, 200), min(L, 500_000)                              # 1e8 cells
        d = sv.synth_multiallelic_coded(Gs, 0, Ls, genome_base=0, seed=args.seed)
        ref = oa.Population("gnomad")
        ref.add_genomes(["Reference"])
        ref.add_records_coded("chr1", d["offsets"], d["ref_code"], d["n_alts"], d["alt_code"], d["af_flat"], None, oa.Population.REFERENCE)
        dip = oa.Population("diploid")
        dip.add_genomes(sv.genome_ids(Gs))
        dip.add_records_coded("chr1", d["offsets"], d["ref_code"], d["n_alts"], d["alt_code"], d["af_flat"], d["alleles"], oa.Population.PHASED)
        counts, freqs, present, seconds = oa.inbreed_window(ref.filter_snp_pass(), dip, np.full(Gs, oa.ALL, dtype=np.int32), "Simple", 0,
                                                            int(d["offsets"][-1]) + 1, 1, 10**9, 0.0, 1.0)
        threads = int(oa.lib().kgo_pool_threads(Gs))
        got = m.inbreed(np.ascontiguousarray(table[:Ls]), "Simple", phased=True, locus_index=np.arange(Ls, dtype=np.uint32), g0=0, g1=Gs)
        got = got[dip.genome_order()]
        names = ["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]
        ok = bool(present.all()) and all(np.array_equal(got[name], counts[:, k]) for k, name in enumerate(names))
        f_err = float(np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max())
        ok = ok and f_err <= 1e-10
        record["cpu_baseline"] = {
            "value": Gs * Ls / seconds, "unit": "genomes·loci/s", "cores": threads, "kind": "port",
            "sample": f"genomes 0..{Gs - 1} x loci 0..{Ls - 1} of the same population ({Gs * Ls:.3g} cells, {seconds:.1f} s): oracle "
                      f"getLocusList x 6 super populations + processResults (generateFrequencies + processSimple, one pool task per genome, "
                      f"{threads} threads = hardware_concurrency() - 1 capped by the genome count; host has {os.cpu_count()} cpus); the sparse "
                      f"store itself is the parser's output and is not timed; parity vs GPU on the slice: class counts "
                      f"{'bit-exact' if ok else 'MISMATCH'}, |dF| max {f_err:.1e}",
            "parity_ok": ok}
    # HallME over the same population (kgx_kernels_hall.h: per-genome moments, one pass over the bytes per class of
    # homozygous cell instead of processHallME's 50): calls from seeded reference start points.
    hall_seed = 4242
    classes = 1 + amax

    def iterative_leg(algorithm, calls):
        """One iterative estimator over the whole population from seeded reference starts: wall ms per call (median), and the
        device time of its parts from the library's own HIP events (frequency sweep, class passes, search) of the last call."""
        start = capi.reference_starts(algorithm, hall_seed, G)
        walls, parts, res = [], {}, None
        for i in range(calls + 1):
            t = time.perf_counter()
            res = m.inbreed_resident(table_dev.data_ptr(), n_sel, amax, algorithm, phased=True, start=start)
            if i >= 1:
                walls.append((time.perf_counter() - t) * 1e3)
        parts = {"frequency_sweep_ms": capi.inbreed_last_sweep_ms(), "class_passes_ms": capi.inbreed_last_moments_ms(),
                 "search_ms": capi.inbreed_last_search_ms(), "path": capi.inbreed_last_path()}
        return float(np.median(walls)), len(walls), res, parts

    def moments_roofline(ms, parts, kernel):
        # The unique bytes of the call are ONE read of the matrix (sweep_bytes).  The call reads it twice: the frequency sweep, then
        # the one pass that leaves every class's hits as bit rows (k_class_bits: a bit per cell of the loci that have the class --
        # all of them for the major class and alt 1, alt 2 and 3 at the share of loci that have them, from the table), which the
        # matrix-core pass reads back.  `achieved` prices the whole call at its unique bytes; `class_passes_GBps` the class passes
        # (bits pass + moment passes + merges) at the bytes they must move: the matrix once, the bit rows out and in.
        has_alt = np.isfinite(table).mean(axis=0)
        cover = 1.0 + float(has_alt.sum())                                   # class rows per locus, on average
        bit_bytes = float(G) * L * cover / 8.0
        class_bytes = float(G) * L + 2.0 * bit_bytes
        return {"bound": "hbm", "kernel": kernel, "achieved": sweep_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": sweep_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": sweep_bytes,
                "what": "the whole call priced at ONE read of the matrix (its unique bytes)",
                "matrix_reads_per_call": 2.0, "bit_rows_bytes": bit_bytes, "class_rows_per_locus": cover,
                "class_passes_bytes": class_bytes,
                "class_passes_GBps": class_bytes / (parts["class_passes_ms"] * 1e-3) / 1e9 if parts["class_passes_ms"] > 0 else None,
                "class_passes_frac": class_bytes / (parts["class_passes_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if parts["class_passes_ms"] > 0 else None,
                "device_ms": parts}

    hall_ms, hall_calls, hall, hall_parts = iterative_leg("HallME", 3)
    record["hallme"] = {
        "metric": "genomes·loci/sec (inbreeding sweep + HallME)", "value": G * L / (hall_ms * 1e-3), "unit": "genomes·loci/s",
        "ms_per_call": hall_ms, "calls": hall_calls,
        "config": {"workload": label, "algorithm": "HallME", "start_points": f"kgx_inbreed_reference_starts(seed {hall_seed})",
                   "passes_over_the_bytes": f"1 frequency sweep + 1 pass that leaves the {classes} classes' hits as bit rows (moments from those by int8 MFMA) instead of 1 + 50",
                   "mean_F": float(hall["inbred_allele_sum"].mean())},
        "roofline": moments_roofline(hall_ms, hall_parts, "k_class_bits (the classes' hits as bit rows), k_hall_mfma<false, true> (moments on the matrix cores), k_hall_iterate"),
        "cpu_baseline": None,
    }
    # Loglikelihood, the reference's DEFAULT estimator (kga_analysis_inbreed_args.h:138), over the same population: the same
    # moments plus the exact walk of the cells next to the 1e-10 floor, the reference optimiser's search run per genome in one
    # workgroup (kgx_kernels_loglik.h) -- instead of 38 passes over the bytes.
    ll_ms, ll_calls, ll, ll_parts = iterative_leg("Loglikelihood", 3)
    record["loglikelihood"] = {
        "metric": "genomes·loci/sec (inbreeding sweep + Loglikelihood)", "value": G * L / (ll_ms * 1e-3), "unit": "genomes·loci/s",
        "ms_per_call": ll_ms, "calls": ll_calls, "evaluations": capi.inbreed_last_evaluations(),
        "config": {"workload": label, "algorithm": "Loglikelihood", "start_points": f"kgx_inbreed_reference_starts(seed {hall_seed})",
                   "passes_over_the_bytes": f"1 frequency sweep + 1 pass that leaves the {classes} classes' hits as bit rows instead of 1 + ~38 (two evaluations each)",
                   "mean_F": float(ll["inbred_allele_sum"].mean())},
        "roofline": moments_roofline(ll_ms, ll_parts, "k_class_bits, k_hall_mfma<true, true> (moments, and the hits' words of the reachable bins), k_loglik_search"),
        "cpu_baseline": None,
    }
    if cpu:
        counts, freqs, present, seconds = oa.inbreed_window(ref.filter_snp_pass(), dip, np.full(Gs, oa.ALL, dtype=np.int32), "HallME", 0,
                                                            int(d["offsets"][-1]) + 1, 1, 10**9, 0.0, 1.0, seed=hall_seed)
        order = dip.genome_order()
        slice_start = np.empty(Gs, dtype=np.float64)
        slice_start[order] = capi.reference_starts("HallME", hall_seed, Gs)       # the k-th genome in id order owns stream seed + k
        got = m.inbreed(np.ascontiguousarray(table[:Ls]), "HallME", phased=True, locus_index=np.arange(Ls, dtype=np.uint32), g0=0, g1=Gs,
                        start=slice_start)[order]
        f_err = float(np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max())
        ok = bool(present.all()) and np.array_equal(got["total_allele_count"], counts[:, 4]) and f_err <= 1e-9
        record["hallme"]["cpu_baseline"] = {
            "value": Gs * Ls / seconds, "unit": "genomes·loci/s", "cores": threads, "kind": "port",
            "sample": f"the Simple leg's slice ({Gs * Ls:.3g} cells, {seconds:.1f} s): oracle processResults with processHallME (5 restarts of 50 "
                      f"steps, the fifth decides; {threads} pool threads), same seeded entropy; parity vs GPU on the slice: |dF| max {f_err:.1e} (bound 1e-9)",
            "parity_ok": ok}
        counts, freqs, present, seconds = oa.inbreed_window(ref.filter_snp_pass(), dip, np.full(Gs, oa.ALL, dtype=np.int32), "Loglikelihood", 0,
                                                            int(d["offsets"][-1]) + 1, 1, 10**9, 0.0, 1.0, seed=hall_seed)
        slice_start[order] = capi.reference_starts("Loglikelihood", hall_seed, Gs)
        got = m.inbreed(np.ascontiguousarray(table[:Ls]), "Loglikelihood", phased=True, locus_index=np.arange(Ls, dtype=np.uint32), g0=0, g1=Gs,
                        start=slice_start)[order]
        slice_path = capi.inbreed_last_path()
        f_abs = np.abs(got["inbred_allele_sum"] - freqs[:, 4])
        f_err, on_path = float(f_abs.max()), int((f_abs <= 2e-6).sum())
        # (one optimiser, one start, each side stopped at a simplex of 1e-6: 2e-6; a genome whose two paths parted at a comparison
        # of values closer than their rounding may sit on a neighbouring maximum -- at most 1 % of them, tests/test_inbreed_gpu.py)
        ok = bool(present.all()) and np.array_equal(got["total_allele_count"], counts[:, 4]) and on_path >= 0.99 * Gs
        record["loglikelihood"]["cpu_baseline"] = {
            "value": Gs * Ls / seconds, "unit": "genomes·loci/s", "cores": threads, "kind": "port",
            "sample": f"the Simple leg's slice ({Gs * Ls:.3g} cells, {seconds:.1f} s): oracle processResults with processLogLikelihood (5 restarts of the "
                      f"1-D Nelder-Mead, the fifth decides; {threads} pool threads), same seeded entropy; parity vs GPU on the slice (path '{slice_path}'): "
                      f"|dF| <= 2e-6 on {on_path} of {Gs} genomes, largest {f_err:.1e}",
            "parity_ok": ok}
    m.close()
    capi.release_scratch()
    return record


def aux_k1_flatten(args, capi):
    """K1 as its own phase (SURVEY.md 8d "build-D"): VariantDBVariant::createVariantDB (kgl_variant_db_variant.cpp:11-123)
    turns the PopulationDB a parser delivered into the dense dosage matrix.  CPU leg: the oracle's createVariantDB on the
    reference's pool (hw - 1 threads, one task per genome).  Product: the GPU_ALLELE package, driven through its
    VirtualAnalysis surface (kgx_host_driver) over the SAME 1000 x 20,000 records, flattens the PopulationDB into 2-bit rows
    on the host and uploads them; it logs both times.  Parity: the package's VariantFWS.csv rows (summaryByVariant of every
    variant, HGVS order) against the oracle's, bit for bit."""
    import re
    import subprocess
    import tempfile

    from tests import oracle_api as oa
    from tests import records_io as rio

    G, V = 1000, 20_000
    rows, af = capi.synth_biallelic_host(args.seed, 0, G, 0, V)
    codes = capi.unpack_dosage2(rows, G)
    rng = np.random.default_rng(7)
    offsets = np.cumsum(rng.integers(1, 51, V)).astype(np.uint64)
    ref_code = rng.integers(0, 4, V).astype(np.uint8)
    alt_code = ((ref_code + rng.integers(1, 4, V)) % 4).astype(np.uint8)
    gt = np.zeros((V, G, 2), dtype=np.uint8)
    het_phase = rng.integers(0, 2, (V, G)).astype(np.uint8)
    gt[..., 0] = np.where(codes == 2, 1, np.where((codes == 1) & (het_phase == 0), 1, 0))
    gt[..., 1] = np.where(codes == 2, 1, np.where((codes == 1) & (het_phase == 1), 1, 0))
    ids = [f"HG{i:06d}" for i in range(G)]
    opop = oa.Population("synthetic")
    opop.add_genomes(ids)
    opop.add_records_coded("chr1", offsets, ref_code, np.ones(V, dtype=np.uint8), alt_code, np.repeat(af[:, None], 6, axis=1), gt, oa.Population.PHASED)
    vdb = oa.VariantDB(opop)                                       # createVariantDB, timed inside the oracle
    want = vdb.summary_by_variant()
    threads = int(oa.lib().kgo_pool_threads(G))
    bases = "ACGT"
    rec = oa.Records("chr1", offsets, [bases[c] for c in ref_code], [[bases[c]] for c in alt_code], af=[np.repeat(a, 6)[None, :] for a in af])
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / "population.bin"
        rio.write_records(path, rec, gt, ids, oa.Population.PHASED, "Genome1000", population_id="synthetic")
        if not rio.DRIVER.exists():
            from kgl_gene_amd import build as kbuild
            kbuild.build_host()
        t0 = time.perf_counter()
        res = subprocess.run([str(rio.DRIVER), "GPU_ALLELE", tmp, "--", str(path)], capture_output=True, text=True,
                             env=dict(os.environ, KGX_FLATTEN_TRACE="1"))
        process_seconds = time.perf_counter() - t0
        log = res.stdout + res.stderr
        m = re.search(r"K1 \(createVariantDB's work\): flatten ([0-9.eE+-]+) s, device rows created and uploaded ([0-9.eE+-]+) s, (\d+) genomes x (\d+) rows", log)
        if res.returncode != 0 or not m:
            return {"error": f"kgx_host_driver GPU_ALLELE failed (rc {res.returncode}): {log[-400:]}"}
        flatten_s, upload_s = float(m.group(1)), float(m.group(2))
        header, rows_csv = rio.read_csv(Path(tmp) / "VariantFWS.csv")
        got = np.array([[int(x) for x in r[-3:]] for r in rows_csv], dtype=np.uint64)
        hgvs_ok = [r[0] for r in rows_csv] == [vdb.hgvs(i) for i in range(vdb.n_variants)]
        ok = bool(hgvs_ok and got.shape == want.shape and np.array_equal(got, want))
    cells = G * V
    return {
        "metric": "variants·genomes/sec (K1: PopulationDB -> dosage rows resident for the sweeps)",
        "value": cells / (flatten_s + upload_s), "unit": "variants·genomes/s", "seconds": {"flatten_host": flatten_s, "create_and_upload": upload_s},
        "config": {"workload": f"{G} genomes x {V} variants of the headline's synthetic population as a PopulationDB ({int(opop.variant_count())} Variant objects)",
                   "boundary": "GPU_ALLELE through kgx_host_driver (VirtualAnalysis::fileReadAnalysis); the PopulationDB itself is the parser's output, untimed on both sides",
                   "whole_process_seconds": round(process_seconds, 3), "rows_on_device": int(m.group(4)),
                   "flatten_phases_s": {what.strip(): float(sec) for what, sec in re.findall(r"kgx flattenPopulation: (.*?) ([0-9.]+) s", log)}},
        "cpu_baseline": {"value": cells / vdb.build_seconds, "unit": "variants·genomes/s", "cores": threads, "kind": "port",
                         "sample": f"the same PopulationDB through the oracle's createVariantDB ({vdb.build_seconds:.3f} s on {threads} pool threads = "
                                   f"hardware_concurrency() - 1 capped by the genome count); parity: the package's VariantFWS.csv == summaryByVariant of every "
                                   f"variant in HGVS order: {'bit-exact' if ok else 'MISMATCH'}",
                         "parity_ok": ok},
    }


def aux_window_calls(args, capi):
    """The regime the INBREED package runs in (kga_analysis_inbreed_diploid.cpp:98-166 per window; defaults LociiCount 1000):
    one kgx_inbreed call per window and super population -- here 1000 sampled loci x 2504 genomes, indexed out of a resident
    20,000-locus matrix, all four estimators, the iterative ones from seeded reference start points.  calls/s from 200 calls
    each; CPU leg: the oracle's getLocusList + processResults over the same window (hw - 1 pool threads), parity per estimator."""
    from tests import oracle_api as oa
    from tests import synth_vcf as sv

    G, L, step, seed = 2504, 20_000, 20, 4242
    m = capi.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(args.seed, 0, 0)
    index = np.arange(0, L, step, dtype=np.uint32)
    sub = np.ascontiguousarray(table[index])
    d = sv.synth_multiallelic_coded(G, 0, L, seed=args.seed)
    first_alt = np.concatenate([[0], np.cumsum(d["n_alts"])]).astype(np.int64)
    keep = np.concatenate([np.arange(first_alt[l], first_alt[l + 1]) for l in index])
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records_coded("chr1", d["offsets"][index], d["ref_code"][index], d["n_alts"][index], d["alt_code"][keep], d["af_flat"][keep], None, oa.Population.REFERENCE)
    dip = oa.Population("diploid")
    dip.add_genomes(sv.genome_ids(G))
    dip.add_records_coded("chr1", d["offsets"][index], d["ref_code"][index], d["n_alts"][index], d["alt_code"][keep], d["af_flat"][keep],
                          np.ascontiguousarray(d["alleles"][index]), oa.Population.PHASED)
    ref_f = ref.filter_snp_pass()
    order = dip.genome_order()
    upper = int(d["offsets"][index[-1]]) + 1
    sp = np.full(G, oa.ALL, dtype=np.int32)
    threads = int(oa.lib().kgo_pool_threads(G))
    for _ in range(100):                                            # clocks up
        m.inbreed(sub, "Simple", phased=True, locus_index=index)
    out = {"metric": "kgx_inbreed calls/sec at window size", "unit": "calls/s",
           "config": {"workload": f"{len(index)} sampled loci (every {step}th of {L}) x {G} genomes per call", "start_points": f"kgx_inbreed_reference_starts(seed {seed})"},
           "algorithms": {}}
    names = ["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]
    tolerance = {"Simple": 1e-10, "RitlandLocus": 1e-10, "HallME": 1e-9, "Loglikelihood": 2e-6}
    for algorithm in ("Simple", "RitlandLocus", "HallME", "Loglikelihood"):
        start = None
        if algorithm in ("HallME", "Loglikelihood"):
            start = np.empty(G)
            start[order] = capi.reference_starts(algorithm, seed, G)
        got = m.inbreed(sub, algorithm, phased=True, locus_index=index, start=start)
        got_first = got.copy()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            m.inbreed(sub, algorithm, phased=True, locus_index=index, start=start)
        per_call = (time.perf_counter() - t0) / reps
        counts, freqs, present, seconds = oa.inbreed_window(ref_f, dip, sp, algorithm, 0, upper, 1, 10**9, 0.0, 1.0, seed=seed)
        got = got[order]
        f_err = float(np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max())
        ok = bool(present.all()) and all(np.array_equal(got[name], counts[:, k]) for k, name in enumerate(names)) and f_err <= tolerance[algorithm]
        out["algorithms"][algorithm] = {
            "value": 1.0 / per_call, "ms_per_call": per_call * 1e3, "genomes_loci_per_s": G * len(index) / per_call,
            "cpu_baseline": {"value": 1.0 / seconds, "unit": "calls/s", "cores": threads, "kind": "port",
                             "sample": f"the same window through the oracle's getLocusList x 6 + processResults ({seconds:.3f} s, {threads} pool threads); "
                                       f"class counts {'bit-exact' if ok else 'MISMATCH or F off'}, |dF| max {f_err:.1e} (bound {tolerance[algorithm]:g})",
                             "parity_ok": ok}}
        if algorithm == "Loglikelihood":
            out["algorithms"][algorithm]["evaluations"] = capi.inbreed_last_evaluations()
        # ... and as the package issues them since round 4: K windows sampled ahead, ONE kgx_inbreed_batch for all of them
        # (one copy in, two launches, one copy out; kgx_kernels_window.h) -- here 16 windows of the same size, shifted by a locus each
        K = 16
        tasks = [{"locus_index": index + k, "minor_af": np.ascontiguousarray(table[index + k]), "start": start} for k in range(K)]
        batch = m.inbreed_batch(tasks, algorithm, phased=True)
        same = all(np.array_equal(batch[0][name], got_first[name]) for name in names) and \
            float(np.nanmax(np.abs(batch[0]["inbred_allele_sum"] - got_first["inbred_allele_sum"]))) <= tolerance[algorithm]
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            m.inbreed_batch(tasks, algorithm, phased=True)
        per_window = (time.perf_counter() - t0) / (reps * K)
        out["algorithms"][algorithm]["batched"] = {"windows_per_batch": K, "ms_per_window": per_window * 1e3, "windows_per_s": 1.0 / per_window,
                                                   "speedup_over_single_calls": per_call / per_window,
                                                   "first_window_equals_the_single_call": bool(same)}
    out["value"] = out["algorithms"]["Simple"]["value"]
    m.close()
    capi.release_scratch()
    return out


def inbreed_workload(args, capi, dist, torch, dev, wl, L, n_gpus, rank):
    """C5: every rank sweeps its own genomes (no collective: per-genome results only need that genome's bytes and the
    per-locus tables).  A step = one kgx_inbreed call = the frequency sweep + the estimator's passes.  The boundary
    takes the per-locus AF table from the host or from device memory; here it is resident in HBM with the genotypes
    before the timed region starts (a host table adds a 120 MB upload to every call: see DESIGN.md); roofline.achieved is
    the frequency-sweep kernel time alone (HIP events inside the library)."""
    G = args.genomes or wl["genomes_per_gpu"]
    m = capi.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(args.seed, rank * G, 0)
    sweep_bytes = int(capi.lib().kgx_gt8_sweep_bytes(G, L, table.shape[1]))

    def fence():
        if n_gpus > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    table_dev = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(dev)
    n_sel, amax = table.shape
    res = None
    for _ in range(args.warmup):
        res = m.inbreed_resident(table_dev.data_ptr(), n_sel, amax, args.algorithm, phased=True)
    fence()
    t0 = time.perf_counter()
    sweep_ms, kernel_ms = [], []
    for _ in range(args.steps):
        res = m.inbreed_resident(table_dev.data_ptr(), n_sel, amax, args.algorithm, phased=True)
        sweep_ms.append(capi.inbreed_last_sweep_ms())
        kernel_ms.append(capi.inbreed_last_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    if n_gpus > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k5_ms = float(np.median(kernel_ms))
    achieved = sweep_bytes / (k5_ms * 1e-3) / 1e9
    if rank != 0:
        return None
    return {
        "metric": "genomes·loci/sec (inbreeding sweep + " + args.algorithm + ")",
        "value": n_gpus * G * L * args.steps / elapsed,
        "unit": "genomes·loci/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 classes, f64 sums", "data": "synthetic",
        "config": {"workload": workload_label(args, wl, G, L), "genomes_per_gpu": G, "loci": L, "algorithm": args.algorithm,
                   "layout": "gt8 allele-index bytes, locus-major", "exchange": "none (genomes are independent)",
                   "mean_F": float(res["inbred_allele_sum"].mean()), "seed": args.seed},
        "roofline": {"bound": "hbm", "kernel": "k_inbreed_eval_lut<3|4> (the frequency sweep's table pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": committed_traffic("K5", workload_label(args, wl, G, L))[0] if args.algorithm != "RitlandLocus" and n_gpus == 1 else None,
                     "traffic_source": (committed_traffic("K5", workload_label(args, wl, G, L))[1] if args.algorithm != "RitlandLocus" and n_gpus == 1
                                        else "no rocprofv3 --pmc pass committed for this kernel"), "algorithmic_bytes_per_launch": sweep_bytes,
                     "kernel_ms": k5_ms, "kernel_ms_statistic": "median", "sweep_ms_with_locus_helpers": float(np.median(sweep_ms))},
        "cpu_baseline": None,
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus:
        if world == 1 and n_gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        n_gpus = world

    import torch
    import torch.distributed as dist

    from kgl_gene_amd import capi
    from kgl_gene_amd.sharding import allreduce_counts_async, replicate_genomes, shard_genomes

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback)")
    # Rehearsal on a 1-GPU box only: KGX_BENCH_REHEARSAL=1 puts every rank on device 0 and exchanges through gloo
    # (RCCL refuses two ranks on one device).  The driver's runs use one GPU per rank and RCCL.
    rehearsal = os.environ.get("KGX_BENCH_REHEARSAL") == "1"
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if n_gpus > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the exchange of batch i runs beside the sweep of batch i+1: give RCCL's stream priority so that its few workgroups
        # are placed as soon as the sweep's (many, short) workgroups free slots
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    # One rank per node builds the extension if it is missing; the others load it only after that rank is done.
    if local_rank == 0:
        capi.ensure_built()
    if n_gpus > 1:
        dist.barrier()
    capi.init(device_index)                                           # this process owns ONE device: one shard per handle

    wl = WORKLOADS[args.workload]
    V = args.variants or wl["variants"]
    if args.workload == "c5":
        result = inbreed_workload(args, capi, dist, torch, dev, wl, V, n_gpus, rank)
        if n_gpus > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result, ensure_ascii=False))
        return
    if args.workload == "c4" and not args.genomes:
        if n_gpus < 2:
            sys.exit("workload c4 (100k x 10M = 250 GB) is sharded: run it with --gpus >= 2")
        shards = shard_genomes(wl["total_genomes"], n_gpus)
        scaling = "strong"
    else:
        shards = replicate_genomes(args.genomes or wl.get("genomes_per_gpu", 10_000), n_gpus)
        scaling = "weak"

    def sweep_job(shards, keep_population):
        """The timed region over one population cut into `shards` (one per rank): W warm-up + K steps, each the K2 sweep of
        this rank's rows, the one exchange of the path (all-reduce of the [V][4] counts, beside the next batch's sweep) and
        the AF epilogue; barrier + synchronize on both sides, MAX over ranks.  Then the dominant kernel alone (HIP events).
        Returns a dict; the population and the last batch's counts stay alive when keep_population."""
        total_genomes = sum(s.n_genomes for s in shards)
        G = shards[rank].n_genomes
        pop = capi.Population(G, V)
        t0 = time.perf_counter()
        pop.synth_biallelic(args.seed, shards[rank].genome_base, 0)
        capi.synchronize()
        t_synth = time.perf_counter() - t0

        # uint32 bit patterns in int32 tensors; sums < 2^31.  With N > 1 two buffers alternate: the all-reduce of batch i
        # (RCCL's own stream, over xGMI) runs beside the sweep of batch i+1, and batch i's AF epilogue follows its sums.
        bufs = [torch.empty((V, 4), dtype=torch.int32, device=dev) for _ in range(2 if n_gpus > 1 else 1)]
        af = torch.empty((V,), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        pending = []                                                       # at most one (work, buffer) in flight
        issued = [0]
        region_events = None                                               # a list while the timed region runs

        def finish():
            while pending:
                work, buf = pending.pop(0)
                if work is not None:
                    work.wait()                                            # the current stream waits; the host does not
                capi.allele_frequency_dev(buf.data_ptr(), V, total_genomes, af.data_ptr(), stream)

        def step():
            buf = bufs[issued[0] % len(bufs)]
            issued[0] += 1
            if region_events is not None:
                # the dominant kernel's launches INSIDE the timed region, HIP events on the stream it is launched on
                # (torch's current stream, so torch.cuda.Event records there)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                pop.allele_count_by_locus_dev(buf.data_ptr(), stream)
                e1.record()
                region_events.append((e0, e1))
            else:
                pop.allele_count_by_locus_dev(buf.data_ptr(), stream)
            work = allreduce_counts_async(buf, n_gpus)                     # the one exchange step of the path
            finish()                                                       # the batch before this one
            pending.append((work, buf))
            if n_gpus == 1:
                finish()

        def fence():
            finish()                                                       # every batch's sums and AF are inside the timed region
            if n_gpus > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(args.warmup):
            step()
        fence()
        region_events = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        region_ms = [e0.elapsed_time(e1) for e0, e1 in region_events]
        region_events = None
        counts = bufs[(issued[0] - 1) % len(bufs)]                        # the last batch's (summed) counts
        # Exchange self-check outside the timed region, on the device, over EVERY variant and on every rank: the four summed
        # counts of a row cover every genome of every rank exactly once; and the AF epilogue read those sums.
        row_ok = counts.sum(dim=1, dtype=torch.int64) == total_genomes
        # (divided by a TENSOR: torch turns a division by a Python scalar into a multiplication by its reciprocal on the
        # device, which rounds differently from the IEEE division the epilogue and the reference make)
        want_af = (counts[:, 1].to(torch.float64) + 2.0 * counts[:, 2].to(torch.float64)) / torch.full((), 2.0 * total_genomes, dtype=torch.float64, device=dev)
        checks = torch.stack([row_ok.all(), (af == want_af).all()]).to(torch.int32)
        del row_ok, want_af
        # per rank: the mean launch of the dominant kernel inside the region, and the region's own wall time -- their spread over the
        # ranks shows a straggler, and (step time - sweep time) what of the exchange and the epilogue is NOT hidden beside the sweeps
        kernel_mean = float(np.mean(region_ms)) if region_ms else 0.0
        spread = {"kernel_ms_min": kernel_mean, "kernel_ms_max": kernel_mean, "step_ms_min": elapsed / args.steps * 1e3, "step_ms_max": elapsed / args.steps * 1e3}
        if n_gpus > 1:
            dist.all_reduce(checks, op=dist.ReduceOp.MIN)                  # every rank must agree
            lo = torch.tensor([kernel_mean, elapsed / args.steps * 1e3], dtype=torch.float64, device=dev)
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            spread = {"kernel_ms_min": float(lo[0]), "kernel_ms_max": float(hi[0]), "step_ms_min": float(lo[1]), "step_ms_max": float(hi[1])}
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        spread["exposed_exchange_and_epilogue_ms"] = spread["step_ms_max"] - spread["kernel_ms_max"]
        exchange_ok, af_ok = (bool(x) for x in checks.tolist())

        # ... and the same kernel alone, back to back into a scratch buffer (a cross-check of the figure above).
        scratch = torch.empty((V, 4), dtype=torch.int32, device=dev)
        ms = pop.allele_count_timed(scratch.data_ptr(), stream, 2, max(args.steps, 10))
        del scratch
        job = dict(total_genomes=total_genomes, G=G, elapsed=elapsed, exchange_ok=exchange_ok, af_ok=af_ok, ms=region_ms, ms_alone=ms, t_synth=t_synth, spread=spread,
                   sweep_bytes=pop.sweep_bytes)                           # V*ceil(G/4) + 16*V
        if keep_population:
            job.update(pop=pop, counts=counts, bufs=bufs, af=af)
        else:
            pop.close()
            del bufs, af, counts
            torch.cuda.empty_cache()
        return job

    def exchange_description():
        if n_gpus == 1:
            return "none (1 GPU)"
        if rehearsal:
            return "gloo rehearsal on one device"
        return "RCCL all-reduce(sum,u32) of [V][4] counts, overlapped with the next batch's sweep"

    def distributed_config(job):
        """What the exchange actually ran on, as torch.distributed and the library report it; and over the ranks the spread of
        the dominant kernel's mean launch and of the step time, with what of a step is not the sweep (exchange + epilogue exposed)."""
        return {"world_size": dist.get_world_size() if n_gpus > 1 else 1,
                "backend": dist.get_backend() if n_gpus > 1 else "none (single process)",
                "kgx_exchange_kind": capi.exchange_kind(),             # inside this process: one device per rank, so "none"
                "kgx_bound_devices": capi.bound_devices(),
                "over_ranks": job["spread"]}

    job = sweep_job(shards, keep_population=True)
    total_genomes, G, elapsed, ms = job["total_genomes"], job["G"], job["elapsed"], job["ms"]
    pop, counts, bufs, af = job["pop"], job["counts"], job["bufs"], job["af"]
    exchange_ok, t_synth, sweep_bytes = job["exchange_ok"], job["t_synth"], job["sweep_bytes"]
    k2_ms = float(np.mean(ms))                                         # the average launch of the timed region
    achieved = sweep_bytes / (k2_ms * 1e-3) / 1e9

    result = None
    if rank == 0:
        value = total_genomes * V * args.steps / elapsed
        label = wl["label"] if args.workload == "c4" and not args.genomes else workload_label(args, wl, G, V)
        traffic, traffic_source = committed_traffic("K2", label)      # the newest profiling round over this workload ...
        if traffic is None:
            traffic, traffic_source = k2_traffic(G, V)                # ... or an earlier one over this shape
        result = {
            "metric": "variants·genomes/sec (allele-freq sweep)",
            "value": value,
            "unit": "variants·genomes/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": label,
                "genomes_per_gpu": G,
                "total_genomes": total_genomes,
                "variants": V,
                "layout": "2-bit dosage rows, variant-major",
                "exchange": exchange_description(),
                "exchange_check": f"row sums == total genomes on all {V} variants, on every rank: " + ("ok" if exchange_ok else "MISMATCH")
                                  + "; AF epilogue == (het + 2 hom) / 2G on all of them: " + ("ok" if job["af_ok"] else "MISMATCH"),
                "distributed": distributed_config(job),
                "scaling_series": ("weak: " + str(G) + " genomes per GPU at every N; this line is its N = " + str(n_gpus) + " point"
                                   if scaling == "weak" else "strong: " + str(total_genomes) + " genomes split over the ranks"),
                "seed": args.seed,
                "synth_seconds": round(t_synth, 3),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_allele_count",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": sweep_bytes,
                "kernel_ms": k2_ms,
                "kernel_ms_statistic": (f"mean of the {len(ms)} launches of the timed region (HIP events on the launch stream around each); "
                                        f"median {float(np.median(ms)):.3f}, min {float(np.min(ms)):.3f}; the kernel alone, back to back into a "
                                        f"scratch buffer after the region: median {float(np.median(job['ms_alone'])):.3f}"),
            },
        }
        dense, nv = None, 0
        if n_gpus == 1 and not args.no_cpu_baseline:
            nv = min(args.cpu_sample_variants, V)
            k2_rows = counts[:nv].cpu().numpy().view(np.uint32)
            result["cpu_baseline"], dense, nv = cpu_baseline(capi, pop, G, V, k2_rows, nv, args.seed)
        else:
            result["cpu_baseline"] = None
        if n_gpus == 1 and args.workload == "c3" and not args.no_aux:
            aux = {"k3_fws_bins": aux_by_genome(capi, torch, pop, counts.view(torch.int32), G, V, dense, nv)}
            aux["k2_genome_mask"] = aux_genome_mask(capi, torch, pop, counts, G, V, sweep_bytes, args.seed)
            del dense
            pop.close()
            job.pop("bufs"); job.pop("counts"); job.pop("af")
            bufs = counts = af = None
            torch.cuda.empty_cache()
            aux["c5_simple"] = aux_inbreeding(args, capi, torch, dev, not args.no_cpu_baseline)
            if not args.no_cpu_baseline:                      # both are comparisons with the CPU path on the same input
                aux["k1_flatten"] = aux_k1_flatten(args, capi)
                aux["window_calls"] = aux_window_calls(args, capi)
            result["aux"] = aux

    pop.close()
    job = pop = counts = bufs = af = None
    torch.cuda.empty_cache()
    # With several ranks the weak-scaled C3 line also carries north_star's own multi-GPU configuration, BASELINE.json
    # configs[3]: 100,000 genomes x 10M SNPs split over the ranks (12,500 per rank at N = 8), the same step -- sweep, ONE
    # all-reduce of the per-variant counts, AF epilogue -- timed the same way.  (Every rank takes part: the job is collective.)
    c4 = WORKLOADS["c4"]
    c4_total = args.c4_genomes or (c4["total_genomes"] if not args.genomes and V == c4["variants"] else 0)
    if n_gpus > 1 and args.workload == "c3" and not args.no_aux and c4_total and (c4_total // n_gpus + 63) // 64 * 16 * V <= 200e9:
        c4_shards = shard_genomes(c4_total, n_gpus)
        cjob = sweep_job(c4_shards, keep_population=False)
        if rank == 0:
            c_ms = float(np.median(cjob["ms"]))
            c_achieved = cjob["sweep_bytes"] / (c_ms * 1e-3) / 1e9
            result.setdefault("aux", {})["c4_strong"] = {
                "metric": "variants·genomes/sec (allele-freq sweep)", "value": cjob["total_genomes"] * V * args.steps / cjob["elapsed"],
                "unit": "variants·genomes/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": cjob["elapsed"] / args.steps * 1e3, "scaling": "strong", "dtype": "u32", "data": "synthetic",
                "config": {"workload": c4["label"] if (c4_total, V) == (c4["total_genomes"], c4["variants"]) else
                                       f"custom: {c4_total} genomes x {V} biallelic SNPs sharded over the ranks",
                           "genomes_per_gpu": cjob["G"], "total_genomes": cjob["total_genomes"], "variants": V,
                           "exchange": exchange_description(),
                           "exchange_check": f"row sums == total genomes on all {V} variants, on every rank: " + ("ok" if cjob["exchange_ok"] else "MISMATCH")
                                             + "; AF epilogue: " + ("ok" if cjob["af_ok"] else "MISMATCH"),
                           "distributed": distributed_config(cjob), "synth_seconds": round(cjob["t_synth"], 3)},
                "roofline": {"bound": "hbm", "kernel": "k_allele_count (rank 0's shard)", "achieved": c_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": c_achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": cjob["sweep_bytes"], "kernel_ms": c_ms},
            }
    if n_gpus > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))


if __name__ == "__main__":
    main()
