"""Parity of the inbreeding kernels (K5/K6/K7, through the C ABI) against the oracle's restatement of
kga_analytic/kga_inbreed.  Integer class counts are bit-exact; fp64 sums and the Simple / RitlandLocus
coefficients agree to 1e-12 relative (only the summation order differs).  HallME and Loglikelihood restart
from random points in the reference (one std::mt19937_64 per genome task) and keep the fifth restart: given the
same entropy -- oracle(seed s) hands the k-th task the stream s + k, the device gets the points
kgx_inbreed_reference_starts draws from those streams -- the GPU agrees to 1e-9 (HallME: 50 expectation steps of
the same map from the same point) and 2e-6 (Loglikelihood: the same simplex walk, each side stopping at a width of
1e-6).  The deterministic mode (no start points: the interval midpoints) is checked to lie inside the envelope of
the seeded runs.  Needs a GPU."""
import numpy as np
import pytest

from . import inbreed_inputs as ii
from . import oracle_api as oa
from . import synth_vcf as sv

pytestmark = pytest.mark.gpu

REL = 1e-12
F_BAND = 2e-4
START_SEED = 77


def seeded_starts(kgx, algorithm, seed, order):
    """Start point of every device genome (caller's order) under the oracle's entropy: the k-th genome in id order --
    processResults' fan-out order -- owns the stream seed + k (oracle/kgo_inbreed.cpp: processResults)."""
    if algorithm not in ("HallME", "Loglikelihood"):
        return None
    start = np.empty(len(order), dtype=np.float64)
    start[order] = kgx.reference_starts(algorithm, seed, len(order))
    return start


def build(G, L, mode, seed):
    rec, gt = sv.multiallelic_block(G, L, rng_seed=seed, missing_af_frac=0.03, dup_records=60)   # repeated records: same-phase pairs, >= 3 variants
    ids = sv.genome_ids(G)
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    dip = sv.oracle_population(rec, gt, ids, mode)
    return rec, gt, ids, ref.filter_snp_pass(), dip


def test_class_frequency_table_is_bit_exact(kgx):
    rng = np.random.default_rng(0)
    n, amax = 5000, 4
    af = rng.uniform(0.0, 0.5, (n, amax)).astype(np.float32).astype(np.float64)
    af[rng.random((n, amax)) < 0.4] = np.nan
    af[:50] *= 3.0                                   # sums over 1: rescale branch / invalid loci
    af[50:60] = np.nan                               # empty vectors
    for F in (0.0, 0.25, -0.5):
        got, valid = kgx.locus_class_frequencies(af, F)
        for l in range(n):
            f = af[l][~np.isnan(af[l])]
            if len(f) == 0:
                assert not valid[l]
                continue
            want = oa.class_frequencies(np.clip(f, 0.0, 1.0), F)       # majorHom, majorHet, minorHom, minorHet
            assert np.array_equal(got[l, 1:], want), (l, F)
            s = 0.0
            for x in np.clip(f, 0.0, 1.0):
                s += x
            assert valid[l] == (not (s - 1.0 > 1e-5))
            assert got[l, 0] == min(max(1.0 - min(max(s, 0.0), 1.0), 0.0), 1.0)


@pytest.mark.parametrize("mode", [oa.Population.PHASED, oa.Population.UNPHASED])
@pytest.mark.parametrize("algorithm,path", [("Simple", "default"), ("RitlandLocus", "default"), ("HallME", "default"), ("Loglikelihood", "default"),
                                            ("HallME", "passes"), ("HallME", "fifty-passes"), ("Loglikelihood", "passes"), ("Loglikelihood", "golden"),
                                            ("Loglikelihood", "compacting"), ("Loglikelihood", "table-passes"),
                                            ("Simple", "generic"), ("RitlandLocus", "generic"), ("HallME", "generic"), ("Loglikelihood", "generic"),
                                            ("Simple", "swar16"), ("HallME", "swar16"), ("Simple", "swar4"), ("RitlandLocus", "no-table"),
                                            ("Simple", "sequential"), ("RitlandLocus", "sequential"), ("Simple", "sequential-swar16"), ("Simple", "sequential-swar4")])
def test_inbreed_window_vs_oracle(kgx, mode, algorithm, path, monkeypatch):
    # every kernel flavour against the same oracle window: the table sweep + window-sized fused iteration (default), the
    # multi-kernel table passes, plain golden section, the generic per-cell kernels, the SWAR sweeps (16 and 4 genomes per
    # lane: what the frequency pass falls back to without the table sweep)
    # ("passes": the one-launch iteration off -- HallME and, of a phased population, Loglikelihood on per-genome moments;
    # "fifty-passes" / "table-passes": their passes over the bytes, 50 resp. two evaluations a pass)
    env = {"passes": {"KGX_K7_NO_WAVE": "1"}, "fifty-passes": {"KGX_K7_NO_WAVE": "1", "KGX_K7_HALL_PASSES": "1"},
           "table-passes": {"KGX_K7_NO_WAVE": "1", "KGX_K7_LL_PASSES": "1"},
           "compacting": {"KGX_K7_NO_WAVE": "1", "KGX_K7_LL_PASSES": "1", "KGX_K7_COMPACT_MIN_GENOMES": "4", "KGX_K7_COMPACT_MIN_CELLS": "1"},
           "golden": {"KGX_K7_NO_WAVE": "1", "KGX_K7_GOLDEN": "1"},
           "generic": {"KGX_K5_GENERIC": "1", "KGX_K7_NO_WAVE": "1"}, "swar16": {"KGX_K5_NO_TABLE_SWEEP": "1"},
           "swar4": {"KGX_K5_NO_TABLE_SWEEP": "1", "KGX_K5_NO_SWAR16": "1"},
           "no-table": {"KGX_K5_NO_EVAL_LUT": "1"},
           # the class-frequency sums of the defaults in the reference's sequential order (the path every call >= 65536 loci takes)
           "sequential": {"KGX_K5_SEQUENTIAL_MIN": "1"}, "sequential-swar16": {"KGX_K5_SEQUENTIAL_MIN": "1", "KGX_K5_NO_TABLE_SWEEP": "1"},
           "sequential-swar4": {"KGX_K5_SEQUENTIAL_MIN": "1", "KGX_K5_NO_TABLE_SWEEP": "1", "KGX_K5_NO_SWAR16": "1"}}.get(path, {})
    for key, value in env.items():
        monkeypatch.setenv(key, value)
    G, L = 101, 1200
    rec, gt, ids, ref, dip = build(G, L, mode, seed=5)
    loci = ii.ReferenceLoci(rec)
    amax = max(len(a) for a in loci.alts)
    bytes_ = ii.encode_gt8(rec, gt, loci, phased_order=(mode == oa.Population.PHASED))
    m = kgx.GenotypeMatrix(G, len(loci.offsets))
    m.load_rows(bytes_)
    assert np.array_equal(m.read_rows(), bytes_)
    order = dip.genome_order()
    lower, upper, spacing, min_af, max_af = 200, 40_000, 60, 0.02, 0.9
    for sp in (oa.ALL, 2):
        table = loci.af_table(sp, amax)
        sel = loci.sample(table, lower, upper, spacing, min_af, max_af)
        want_offsets = oa.sample_locii(ref, sp, False, lower, upper, spacing, 1000, min_af, max_af)
        assert np.array_equal(loci.offsets[sel], want_offsets)
        counts, freqs, present, _ = oa.inbreed_window(ref, dip, np.full(G, sp, dtype=np.int32), algorithm, lower, upper,
                                                      spacing, 1000, min_af, max_af, seed=START_SEED)
        assert present.all()
        got = m.inbreed(table[sel], algorithm, phased=(mode == oa.Population.PHASED), locus_index=sel,
                        start=seeded_starts(kgx, algorithm, START_SEED, order))[order]
        if algorithm == "Loglikelihood" and path in ("passes", "table-passes", "compacting"):      # (what ran is what the parameter says)
            on_moments = path == "passes" and mode == oa.Population.PHASED
            assert kgx.inbreed_last_path() in (("loglik moments", "loglik moments + passes") if on_moments else ("loglik passes",)), kgx.inbreed_last_path()
        if algorithm == "HallME" and path in ("passes", "fifty-passes"):
            assert kgx.inbreed_last_path() == ("hall moments" if path == "passes" else "hall passes")
        # oracle columns: major_het, minor_het, minor_hom, major_hom, total
        for k, name in enumerate(["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]):
            assert np.array_equal(got[name], counts[:, k]), name
        for k, name in enumerate(["major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"]):
            assert np.allclose(got[name], freqs[:, k], rtol=REL, atol=REL), name
        if algorithm in ("Simple", "RitlandLocus"):
            assert np.allclose(got["inbred_allele_sum"], freqs[:, 4], rtol=1e-10, atol=1e-12)
        elif algorithm == "HallME":
            assert np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max() <= 1e-9
        elif path == "golden":
            # another search (golden section over [-1, 1], no start point): the same maximum within the reference's band
            assert np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max() <= F_BAND
        else:
            assert np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max() <= 2e-6
        assert counts[:, 1].sum() > 0 and counts[:, 2].sum() + counts[:, 1].sum() > 0
    m.close()


def test_midpoint_starts_lie_inside_the_envelope_of_seeded_runs(kgx):
    """kgx_inbreed without start points runs from the midpoints of the reference's start intervals -- a mode the
    reference does not have.  What it returns must lie inside what the reference's own behaviour spans: the min-max
    envelope, per genome, of 16 oracle runs with different entropy.  (HallME's 50-step map is monotone in its start, so
    the midpoint run lies between a run started below 0.25 and one started above; Loglikelihood ends on one of the
    local maxima the seeded runs end on.)  The envelope's width -- the reference's run-to-run scatter -- is printed."""
    G, L = 101, 1200
    rec, gt, ids, ref, dip = build(G, L, oa.Population.PHASED, seed=5)
    loci = ii.ReferenceLoci(rec)
    amax = max(len(a) for a in loci.alts)
    m = kgx.GenotypeMatrix(G, len(loci.offsets))
    m.load_rows(ii.encode_gt8(rec, gt, loci, phased_order=True))
    order = dip.genome_order()
    lower, upper, spacing, min_af, max_af = 200, 40_000, 60, 0.02, 0.9
    table = loci.af_table(oa.ALL, amax)
    sel = loci.sample(table, lower, upper, spacing, min_af, max_af)
    sp = np.full(G, oa.ALL, dtype=np.int32)
    for algorithm, slack in (("HallME", 1e-9), ("Loglikelihood", 2e-6)):
        runs = np.stack([oa.inbreed_window(ref, dip, sp, algorithm, lower, upper, spacing, 1000, min_af, max_af, seed=1000 + 977 * k)[1][:, 4]
                         for k in range(16)])
        lo, hi = runs.min(axis=0), runs.max(axis=0)
        mid = m.inbreed(table[sel], algorithm, phased=True, locus_index=sel)[order]["inbred_allele_sum"]
        print(f"{algorithm}: envelope of 16 seeded oracle runs up to {float((hi - lo).max()):.3g} wide (median {float(np.median(hi - lo)):.3g})")
        outside = (mid < lo - slack) | (mid > hi + slack)
        assert not outside.any(), (algorithm, np.flatnonzero(outside).tolist(), mid[outside].tolist(), lo[outside].tolist(), hi[outside].tolist())
    m.close()


def test_start_points_are_range_checked(kgx):
    m = kgx.GenotypeMatrix(8, 16)
    m.load_rows(np.zeros((16, 8), dtype=np.uint8))
    af = np.full((16, 1), 0.2)
    with pytest.raises(Exception):
        m.inbreed(af, "HallME", phased=True, start=np.full(8, 0.0))          # the reference's draws lie in (0, 0.5]
    with pytest.raises(Exception):
        m.inbreed(af, "Loglikelihood", phased=True, start=np.full(8, 1.5))   # outside the optimiser's box
    m.inbreed(af, "Simple", phased=True, start=np.full(8, 9.0))              # ignored by the closed-form estimators
    m.close()


def test_inbreed_takes_a_device_resident_af_table(kgx):
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")          # the runtime libkgx.so itself is linked against (already loaded)
    G, L = 90, 400
    rng = np.random.default_rng(3)
    rows = rng.choice(np.array([0, 0, 0, 1, 0x11, 0x21, 2, 0x12], dtype=np.uint8), size=(L, G))
    af = rng.uniform(0.01, 0.3, (L, 2))
    m = kgx.GenotypeMatrix(G, L)
    m.load_rows(rows)
    af_dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(af_dev), C.c_size_t(af.nbytes)) == 0
    assert hip.hipMemcpy(af_dev, C.c_void_p(af.ctypes.data), C.c_size_t(af.nbytes), 1) == 0      # hipMemcpyHostToDevice
    try:
        for algorithm in ("Simple", "Loglikelihood"):
            host = m.inbreed(af, algorithm, phased=True)
            resident = m.inbreed_resident(af_dev.value, L, 2, algorithm, phased=True)
            assert host.tobytes() == resident.tobytes(), algorithm
    finally:
        hip.hipFree(af_dev)
        m.close()


def test_genome_major_loader_and_subranges(kgx):
    G, L = 75, 300
    rng = np.random.default_rng(1)
    by_genome = rng.choice(np.array([0, 0, 0, 1, 0x11, 0x21, 2, 0x12, 0xFF, 0x1F], dtype=np.uint8), size=(G, L))
    m = kgx.GenotypeMatrix(G, L)
    m.load_genomes(by_genome[:40], 0)
    m.load_genomes(by_genome[40:], 40)
    assert np.array_equal(m.read_rows(), by_genome.T)
    af = rng.uniform(0.01, 0.3, (L, 2))
    whole = m.inbreed(af, "Simple", phased=True)
    part = m.inbreed(af, "Simple", phased=True, g0=32, g1=64)
    assert np.array_equal(whole[32:64], part)
    m.close()


def test_device_multiallelic_generator_and_oracle_parity(kgx):
    """The C5-shaped synthetic population: device generator == host twin, and the sweep on it == oracle."""
    G, L = 160, 900
    rec, alleles, gt8_host, table_host = sv.synth_multiallelic_block(G, L)
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    assert np.array_equal(m.read_rows(), gt8_host)
    assert np.array_equal(np.isnan(table), np.isnan(table_host)) and np.array_equal(np.nan_to_num(table), np.nan_to_num(table_host))
    ids = sv.genome_ids(G)
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    dip = sv.oracle_population(rec, alleles, ids, oa.Population.PHASED)
    # every locus with a SNP alt, no spacing: the whole matrix is one window
    lower, upper = 0, int(rec.offsets[-1]) + 1
    counts, freqs, present, _ = oa.inbreed_window(ref.filter_snp_pass(), dip, np.full(G, oa.ALL, dtype=np.int32), "Simple",
                                                  lower, upper, 1, 10**6, 0.0, 1.0)
    got = m.inbreed(table, "Simple", phased=True)[dip.genome_order()]
    for k, name in enumerate(["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]):
        assert np.array_equal(got[name], counts[:, k]), name
    assert np.allclose(got["inbred_allele_sum"], freqs[:, 4], rtol=1e-10, atol=1e-12)
    # the estimator recovers the generating inbreeding coefficients on average
    F_true = (np.arange(G) % 101 - 50) / 100.0
    assert np.corrcoef(F_true, got["inbred_allele_sum"][np.argsort(dip.genome_order())])[0, 1] > 0.8
    m.close()


def test_c5_full_size_properties(kgx):
    """BASELINE config 4 on one GPU: 10k genomes x 5M mixed SNP+indel multi-allelic loci (50 GB)."""
    G, L = 10_000, 5_000_000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    res = m.inbreed(table, "Simple", phased=True)
    n_valid = int((~np.isnan(table)).any(1).sum())
    assert np.all(res["total_allele_count"] <= n_valid) and np.all(res["total_allele_count"] > 0.98 * n_valid)
    tot = res["major_hetero_count"] + res["minor_hetero_count"] + res["minor_homo_count"] + res["major_homo_count"]
    assert np.array_equal(tot, res["total_allele_count"])
    fsum = res["major_hetero_freq"] + res["minor_hetero_freq"] + res["minor_homo_freq"] + res["major_homo_freq"]
    assert np.allclose(fsum, res["total_allele_count"], rtol=1e-9)          # class frequencies sum to 1 per classified locus
    F_true = (np.arange(G) % 101 - 50) / 100.0
    assert np.abs(res["inbred_allele_sum"] - F_true).max() < 0.05           # 5M loci pin F tightly
    # sub-block regenerated by the host twin
    l0 = 1_234_567
    host_gt8, host_table, _ = kgx.synth_multiallelic_host(1111, 0, G, l0, l0 + 4)
    assert np.array_equal(m.read_rows(l0, l0 + 4), host_gt8)
    # a genome sub-range gives the same rows as the whole
    part = m.inbreed(table, "Simple", phased=True, g0=4000, g1=4100)
    for name in part.dtype.names:       # integer fields bit-exact; fp64 sums differ only by the segment grouping
        if name.endswith("_count"):
            assert np.array_equal(part[name], res[name][4000:4100])
        else:
            assert np.allclose(part[name], res[name][4000:4100], rtol=1e-11, atol=0)
    m.close()


def _f_grid():
    # the reference's grid, built the way it builds it (kga_analysis_inbreed_syngen.cpp:44-56)
    grid, f = [], -0.5
    while f <= 0.5 + 0.000001:
        grid.append(f)
        f += 0.01
    return np.array(grid, dtype=np.float64)


def test_synth_inbred_draws_follow_class_frequencies(kgx):
    """kgx_gt8_synth_inbred (InbreedSynthetic::generateSyntheticPopulation on the device): every genome's class
    counts sit within sampling error of the sum of the oracle's class frequencies at that genome's F, the allele
    picks follow the minor frequencies, and the estimators recover F (the reference's self-check)."""
    rng = np.random.default_rng(23)
    L, amax = 30000, 3
    table = np.full((L, amax), np.nan)
    table[:, 0] = rng.uniform(0.05, 0.4, L)
    two = rng.random(L) < 0.4
    table[two, 1] = rng.uniform(0.02, 0.2, int(two.sum()))
    three = two & (rng.random(L) < 0.3)
    table[three, 2] = rng.uniform(0.01, 0.1, int(three.sum()))
    grid = _f_grid()
    assert len(grid) == 101
    gm = kgx.GenotypeMatrix(len(grid), L)
    gm.synth_inbred(table, grid, seed=77)
    rows = gm.read_rows()[:, :len(grid)]
    a1, a2 = (rows & 0xF).astype(np.int64), (rows >> 4).astype(np.int64)
    assert a1.max() <= amax and a2.max() <= amax
    # no pick of an allele the locus does not have
    n_alt = np.sum(~np.isnan(table), axis=1)
    assert np.all(a1 <= n_alt[:, None]) and np.all(a2 <= n_alt[:, None])
    cls = {
        "major_hom": (a1 == 0) & (a2 == 0),
        "major_het": (a1 == 0) != (a2 == 0),
        "minor_hom": (a1 != 0) & (a1 == a2),
        "minor_het": (a1 != 0) & (a2 != 0) & (a1 != a2),
    }
    order = ["major_hom", "major_het", "minor_hom", "minor_het"]
    for g in range(0, len(grid), 10):
        freqs = np.array([oa.class_frequencies(table[l][~np.isnan(table[l])], grid[g]) for l in range(0, L, 15)])
        sub = slice(0, L, 15)
        n = freqs.shape[0]
        expect = freqs.sum(axis=0)
        sigma = np.sqrt((freqs * (1 - freqs)).sum(axis=0)) + 1.0
        got = np.array([cls[k][sub, g].sum() for k in order], dtype=np.float64)
        assert got.sum() == n
        assert np.all(np.abs(got - expect) < 5 * sigma), (g, grid[g], got, expect)
    # a single carrier sits in the low nibble (gt8 has no phase slot for it)
    assert not np.any((a1 == 0) & (a2 != 0))
    # the reference's self-check through the device estimators
    for algorithm, tol in (("Simple", 0.05), ("Loglikelihood", 0.05)):
        res = gm.inbreed(table, algorithm, phased=True)
        calc = res["inbred_allele_sum"]
        slope, intercept = np.polyfit(grid, calc, 1)
        assert slope > 0.9 and abs(intercept) < 0.02, (algorithm, slope, intercept)
        mild = grid > -0.2          # strongly negative F clamps the minor-homozygous class at rare loci, which biases every estimator
        assert np.abs(calc - grid)[mild].max() < tol, (algorithm, np.abs(calc - grid)[mild].max())
        assert np.abs(calc - grid).max() < 0.15
    # determinism and seed sensitivity
    gm2 = kgx.GenotypeMatrix(len(grid), L)
    gm2.synth_inbred(table, grid, seed=77)
    assert np.array_equal(gm2.read_rows(), gm.read_rows())
    gm2.synth_inbred(table, grid, seed=78)
    assert not np.array_equal(gm2.read_rows(), gm.read_rows())


@pytest.mark.parametrize("algorithm", ["Simple", "RitlandLocus", "HallME", "Loglikelihood"])
@pytest.mark.parametrize("amax", [3, 6])
def test_inbreed_survives_bytes_no_flattener_writes(kgx, algorithm, amax):
    """Every one of the 256 byte values, valid or not, in every genome column: bytes the layout does not define
    (indices past the table, unknown alts) must be skipped by every sweep flavour, never read through, and the (0, a)
    byte counts as the same-phase pair it stands for.  Checked against a numpy restatement of generateFrequencies'
    class decision."""
    L, G = 512, 64
    rng = np.random.default_rng(amax)
    table = np.full((L, amax), np.nan)
    table[:, 0] = rng.uniform(0.05, 0.3, L)
    table[::2, 1] = rng.uniform(0.02, 0.2, L // 2)
    rows = np.zeros((L, G), dtype=np.uint8)
    for l in range(L):
        rows[l] = (np.arange(G) * 4 + l * 7 + (np.arange(G) % 4)) & 0xFF     # all 256 values appear in every column
    gm = kgx.GenotypeMatrix(G, L)
    gm.load_rows(rows)
    res = gm.inbreed(table, algorithm, phased=True)
    a1, a2 = (rows & 0xF).astype(np.int64), (rows >> 4).astype(np.int64)
    n_alt = amax
    in1 = np.zeros_like(rows, dtype=bool)
    in2 = np.zeros_like(rows, dtype=bool)
    for a in range(1, n_alt + 1):
        has = ~np.isnan(table[:, a - 1])[:, None]
        in1 |= (a1 == a) & has
        in2 |= (a2 == a) & has
    bad = (a1 == 15) | (a2 == 15) | (a1 > n_alt) | (a2 > n_alt)
    major_hom = rows == 0
    major_het = ~bad & in1 & (a2 == 0)
    minor_hom = ~bad & in1 & (a2 != 0) & (a1 == a2)
    # (0, a): two copies of alt a on one phase -> a minor heterozygote with that allele twice
    minor_het = (~bad & in1 & (a2 != 0) & (a1 != a2) & in2) | (~bad & (a1 == 0) & in2)
    assert np.array_equal(res["major_homo_count"], major_hom.sum(axis=0).astype(np.uint64))
    assert np.array_equal(res["major_hetero_count"], major_het.sum(axis=0).astype(np.uint64))
    assert np.array_equal(res["minor_homo_count"], minor_hom.sum(axis=0).astype(np.uint64))
    assert np.array_equal(res["minor_hetero_count"], minor_het.sum(axis=0).astype(np.uint64))
    assert np.all(np.isfinite(res["inbred_allele_sum"]))


def test_kernel_flavours_agree_on_random_shapes(kgx, monkeypatch):
    """Differential fuzz: random shapes (genome counts off every lane width, sub-ranges, indexed and dense loci, 1..14
    alleles, every byte value incl. the (0, a) pair, 0xFF and unknown alts, loci without defaults, missing AFs) through
    every kernel flavour -- the generic per-cell kernels are the ones pinned to the oracle above; the SWAR sweeps, the
    table passes, the moments of HallME and Loglikelihood ("passes": any call size with the one-launch iteration off; "fifty-passes":
    their passes over the bytes) and the one-launch iteration must reproduce them."""
    import os

    rng = np.random.default_rng(int(os.environ.get("KGX_FUZZ_SEED", "2024")))      # other seeds / more trials: a longer hunt, by hand
    flavours = {"default": {}, "generic": {"KGX_K5_GENERIC": "1", "KGX_K7_NO_WAVE": "1"}, "passes": {"KGX_K7_NO_WAVE": "1"},
                "fifty-passes": {"KGX_K7_NO_WAVE": "1", "KGX_K7_HALL_PASSES": "1", "KGX_K7_LL_PASSES": "1"},
                "swar16": {"KGX_K5_NO_TABLE_SWEEP": "1"}, "swar4": {"KGX_K5_NO_TABLE_SWEEP": "1", "KGX_K5_NO_SWAR16": "1"},
                "sequential": {"KGX_K5_SEQUENTIAL_MIN": "1"}}
    knobs = sorted({k for env in flavours.values() for k in env})
    for trial in range(int(os.environ.get("KGX_FUZZ_TRIALS", "150"))):
        G = int(rng.choice([1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 100, 257, 1000]))
        L = int(rng.choice([1, 7, 8, 9, 63, 64, 65, 500, 3000]))
        amax = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 14]))
        phased = bool(rng.integers(0, 2))
        table = np.full((L, amax), np.nan)
        n_alt = rng.integers(1, amax + 1, L)
        for l in range(L):
            f = rng.uniform(0.001, 0.6 / n_alt[l], n_alt[l])
            if rng.random() < 0.03:
                f *= 4.0                                     # sums past 1: invalid / rescaled loci
            if rng.random() < 0.02:
                f[:] = 0.999 / n_alt[l]                      # p_major <= 0.01: a locus without defaults
            table[l, :n_alt[l]] = f
            if rng.random() < 0.05:
                table[l, rng.integers(0, n_alt[l])] = np.nan  # an alt without AF
        values = np.array([0] * 12 + [1, 1, 1, 2, 3, 0x11, 0x21, 0x12, 0x22, 0x10, 0x20, 0x1F, 0xF1, 0xFF, 0x31, 0x4, 0x44, 0x80, 0x08, 0xE1],
                          dtype=np.uint8)
        rows = rng.choice(values, size=(L, G))
        if rng.random() < 0.3:
            rows = rng.integers(0, 256, size=(L, G), dtype=np.uint8)
        m = kgx.GenotypeMatrix(G, L)
        m.load_rows(rows)
        g0 = int(rng.choice([0, 4, 16])) if G > 20 else 0
        g1 = int(rng.integers(g0 + 1, G + 1))
        index = None
        sub = table
        if L > 8 and rng.random() < 0.5:
            index = np.sort(rng.choice(L, int(rng.integers(1, L)), replace=False)).astype(np.uint32)
            sub = np.ascontiguousarray(table[index])
        for algorithm, tol in (("Simple", 1e-9), ("RitlandLocus", 1e-9), ("HallME", 1e-8), ("Loglikelihood", 2e-5)):
            results = {}
            for name, env in flavours.items():
                for k in knobs:
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                results[name] = m.inbreed(sub, algorithm, phased=phased, locus_index=index, g0=g0, g1=g1)
            ref = results["generic"]
            for name, got in results.items():
                ctx = (trial, G, L, amax, phased, algorithm, name, g0, g1, None if index is None else len(index))
                for field in ("major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"):
                    assert np.array_equal(got[field], ref[field]), ctx + (field,)
                for field in ("major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"):
                    assert np.allclose(got[field], ref[field], rtol=1e-11, atol=1e-11), ctx + (field,)
                a, b = got["inbred_allele_sum"], ref["inbred_allele_sum"]
                both = np.isfinite(a) & np.isfinite(b)
                # finiteness may differ only at a 0/0: Simple's denominator N - expected homozygotes, Ritland's / Hall's
                # zero counts -- where the last bit of a sum decides between nan and a number
                odd = np.isfinite(a) != np.isfinite(b)
                if odd.any():
                    n = ref["total_allele_count"][odd].astype(np.float64)
                    het = (ref["major_hetero_freq"] + ref["minor_hetero_freq"])[odd]
                    assert algorithm == "Simple" and np.all(np.abs(het) <= 1e-9 * np.maximum(n, 1.0)), ctx
                if algorithm == "Loglikelihood":
                    continue        # several local maxima on adversarial bytes: compared on realistic data above
                assert np.all(np.abs(a[both] - b[both]) <= tol * np.maximum(1.0, np.abs(b[both]))), ctx
        m.close()


def test_loglikelihood_where_the_upper_clamp_binds(kgx, monkeypatch):
    """Two minor alleles of 0.500004 each pass checkValidAlleleVector (sum <= 1 + 1e-5) and make 2(1-F)f1f2 exceed 1 next
    to F = -1: the one place the upper bound of logLikelihood's clamp (_calc.cpp:117-121) binds.  The table pass applies
    it only in batches whose table can get there; the generic kernel clamps every cell as the reference does."""
    rng = np.random.default_rng(8)
    G, L = 70, 900
    table = np.full((L, 2), np.nan)
    table[:, 0] = rng.uniform(0.05, 0.4, L)
    corner = rng.random(L) < 0.4
    table[corner, 0] = table[corner, 1] = 0.500004
    rows = rng.choice(np.array([0, 0, 1, 0x11, 0x21], dtype=np.uint8), size=(L, G))
    rows[~corner] = np.where(rows[~corner] == 0x21, 1, rows[~corner])   # one alt only elsewhere
    rows[:, :20] = np.where(corner[:, None], 0x21, 1)                   # genomes 0..19: heterozygous everywhere -> the maximum is at F = -1
    m = kgx.GenotypeMatrix(G, L)
    m.load_rows(rows)
    results = {}
    paths = {}
    for name, env in {"default": {}, "passes": {"KGX_K7_NO_WAVE": "1", "KGX_K7_LL_PASSES": "1"}, "moments": {"KGX_K7_NO_WAVE": "1"},
                      "generic": {"KGX_K5_GENERIC": "1", "KGX_K7_NO_WAVE": "1"}}.items():
        for k in ("KGX_K7_NO_WAVE", "KGX_K5_GENERIC", "KGX_K7_LL_PASSES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        results[name] = m.inbreed(table, "Loglikelihood", phased=True)["inbred_allele_sum"]
        paths[name] = kgx.inbreed_last_path()
    assert results["generic"][:20].max() < -0.9                          # the search did go where the clamp binds
    # (on the moments such a cell -- 2*f1*f2 > 1/2 -- sends its genome to the passes: kOddBigHet)
    assert paths["moments"] == "loglik moments + passes" and paths["passes"] == "loglik passes", paths
    for name in ("default", "passes", "moments"):
        assert np.abs(results[name] - results["generic"]).max() <= 2e-5, name
    m.close()


def test_loglikelihood_compaction_is_bit_identical(kgx, monkeypatch):
    """The multi-kernel Loglikelihood search drops finished genomes from its passes (compacted columns and states).  A
    genome's sums do not depend on its neighbours: with and without compaction the coefficients are the same bits."""
    rng = np.random.default_rng(8)
    G, L = 3000, 6000
    table = np.full((L, 3), np.nan)
    table[:, 0] = rng.uniform(0.02, 0.45, L).astype(np.float32)
    two = rng.random(L) < 0.3
    table[two, 1] = rng.uniform(0.01, 0.2, int(two.sum())).astype(np.float32)
    F = rng.uniform(-0.3, 0.3, G)
    m = kgx.GenotypeMatrix(G, L)
    m.synth_inbred(table, F, seed=3)
    index = np.sort(rng.choice(L, 4000, replace=False)).astype(np.uint32)
    sub = np.ascontiguousarray(table[index])
    monkeypatch.setenv("KGX_K7_NO_WAVE", "1")
    monkeypatch.setenv("KGX_K7_LL_PASSES", "1")
    results = {}
    for name, env in (("plain", {"KGX_K7_NO_COMPACT": "1"}), ("compacting", {"KGX_K7_COMPACT_MIN_GENOMES": "64", "KGX_K7_COMPACT_MIN_CELLS": "1000"})):
        for k in ("KGX_K7_NO_COMPACT", "KGX_K7_COMPACT_MIN_GENOMES", "KGX_K7_COMPACT_MIN_CELLS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        results[name] = (m.inbreed(sub, "Loglikelihood", phased=True, locus_index=index, g0=16, g1=G - 7), kgx.inbreed_last_evaluations())
    (a, evals_a), (b, evals_b) = results["plain"], results["compacting"]
    assert evals_a == evals_b and evals_a > 12
    assert np.array_equal(a["inbred_allele_sum"], b["inbred_allele_sum"])
    assert np.array_equal(a["total_allele_count"], b["total_allele_count"])
    assert np.abs(a["inbred_allele_sum"] - F[16:G - 7]).max() < 0.25 and np.median(np.abs(a["inbred_allele_sum"] - F[16:G - 7])) < 0.03
    m.close()


def _reference_population(d):
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records_coded("chr1", d["offsets"], d["ref_code"], d["n_alts"], d["alt_code"], d["af_flat"], None, oa.Population.REFERENCE)
    return ref


def test_c5_full_size_fp64_sums_vs_oracle(kgx):
    """BASELINE config 4 at full size against the oracle, not only against its own identities: the device sweeps 10k
    genomes x 5M multi-allelic loci; a 64-genome slice of the same population (regenerated by the host twin as VCF-like
    records + GT pairs) goes through the oracle.  Class counts bit-exact; the four class-frequency sums, Simple and
    RitlandLocus within 1e-12 * max(1, |x|) of the oracle's sequential sums over all 5M loci.
    (a) all 5M loci x 64 genomes through the oracle's dense tier (bit-identical to generateFrequencies where both
    apply: tests/test_oracle_pins.py); (b) the first 1M loci x 16 genomes through generateFrequencies itself."""
    G, L = 10_000, 5_000_000
    g0, n_slice = 4032, 64                                      # a slice off the 128-genome unit boundary, mixed F
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    simple = m.inbreed(table, "Simple", phased=True)
    ritland = m.inbreed(table, "RitlandLocus", phased=True)
    for name in simple.dtype.names:
        if name != "inbred_allele_sum":
            assert np.array_equal(simple[name], ritland[name]), name            # one frequency sweep, whatever the estimator

    def close(got, want, what):
        err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        assert err.max() <= 1e-12, (what, float(err.max()))

    count_fields = ["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]
    freq_fields = ["major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"]

    # (a) dense tier, every locus
    d = sv.synth_multiallelic_coded(n_slice, 0, L, genome_base=g0)
    assert np.array_equal(m.read_rows(77_000, 77_064)[:, g0:g0 + n_slice], d["gt8"][77_000:77_064])
    ref = _reference_population(d)
    ref_snp = ref.filter_snp_pass()
    upper = int(d["offsets"][-1]) + 1
    counts, freqs, _ = oa.inbreed_dense(ref, ref_snp, oa.ALL, 0, upper, 1, 0.0, 1.0, d["offsets"], d["alleles"], phased=True)
    for k, name in enumerate(count_fields):
        assert np.array_equal(simple[name][g0:g0 + n_slice], counts[:, k]), name
    for k, name in enumerate(freq_fields):
        close(simple[name][g0:g0 + n_slice], freqs[:, k], name)
    close(simple["inbred_allele_sum"][g0:g0 + n_slice], freqs[:, 4], "Simple")
    close(ritland["inbred_allele_sum"][g0:g0 + n_slice], freqs[:, 5], "RitlandLocus")
    assert counts[:, 4].min() > 4_000_000

    # (b) generateFrequencies itself (the pointer-chasing store) on the first 1M loci x 16 genomes of the slice
    L1, n1 = 1_000_000, 16
    index = np.arange(L1, dtype=np.uint32)
    sub_table = np.ascontiguousarray(table[:L1])
    d1 = {k: (v[:L1] if k in ("offsets", "ref_code", "n_alts", "alleles", "gt8", "table") else v) for k, v in d.items()}
    n_flat = int(d["n_alts"][:L1].sum())
    d1["alt_code"], d1["af_flat"] = d["alt_code"][:n_flat], d["af_flat"][:n_flat]
    ref1 = _reference_population(d1).filter_snp_pass()
    dip = oa.Population("diploid")
    dip.add_genomes(sv.genome_ids(n1))
    dip.add_records_coded("chr1", d1["offsets"], d1["ref_code"], d1["n_alts"], d1["alt_code"], d1["af_flat"],
                          np.ascontiguousarray(d1["alleles"][:, :n1]), oa.Population.PHASED)
    upper1 = int(d1["offsets"][-1]) + 1
    for algorithm in ("Simple", "RitlandLocus"):
        c1, f1, present, _ = oa.inbreed_window(ref1, dip, np.full(n1, oa.ALL, dtype=np.int32), algorithm, 0, upper1, 1, 10**9, 0.0, 1.0)
        assert present.all()
        got = m.inbreed(sub_table, algorithm, phased=True, locus_index=index, g0=g0, g1=g0 + n1)[dip.genome_order()]
        for k, name in enumerate(count_fields):
            assert np.array_equal(got[name], c1[:, k]), (algorithm, name)
        for k, name in enumerate(freq_fields):
            close(got[name], f1[:, k], (algorithm, name))
        close(got["inbred_allele_sum"], f1[:, 4], algorithm)
    m.close()


def test_iterative_estimators_at_scale_vs_oracle(kgx):
    """HallME and Loglikelihood on per-genome moments (the path C5 takes) at 1,000 genomes x 100,000 loci
    of the C5 population against the oracle under the same entropy (seeded per-genome streams, the fifth draw starts the
    run that counts): class counts bit-exact, HallME within 1e-9, Loglikelihood within 2e-6 (one optimiser, one start,
    each side converged to a simplex of 1e-6)."""
    G, L = 1000, 100_000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    d = sv.synth_multiallelic_coded(G, 0, L)
    assert np.array_equal(m.read_rows(5000, 5016), d["gt8"][5000:5016])
    ref = _reference_population(d).filter_snp_pass()
    dip = oa.Population("diploid")
    dip.add_genomes(sv.genome_ids(G))
    dip.add_records_coded("chr1", d["offsets"], d["ref_code"], d["n_alts"], d["alt_code"], d["af_flat"], d["alleles"], oa.Population.PHASED)
    upper = int(d["offsets"][-1]) + 1
    order = dip.genome_order()
    sp = np.full(G, oa.ALL, dtype=np.int32)
    f_true = ((np.arange(G) % 101) - 50) / 100.0
    for algorithm in ("HallME", "Loglikelihood"):
        counts, freqs, present, _ = oa.inbreed_window(ref, dip, sp, algorithm, 0, upper, 1, 10**9, 0.0, 1.0, seed=START_SEED)
        assert present.all()
        got = m.inbreed(table, algorithm, phased=True, start=seeded_starts(kgx, algorithm, START_SEED, order))[order]
        assert kgx.inbreed_last_path() == ("hall moments" if algorithm == "HallME" else "loglik moments"), kgx.inbreed_last_path()
        assert np.array_equal(got["total_allele_count"], counts[:, 4])
        assert np.array_equal(got["minor_homo_count"], counts[:, 2]) and np.array_equal(got["major_homo_count"], counts[:, 3])
        err = np.abs(got["inbred_allele_sum"] - freqs[:, 4])
        if algorithm == "HallME":
            assert err.max() <= 1e-9, (algorithm, float(err.max()), int(err.argmax()))
            continue
        # Loglikelihood.  For a genome with F < 0 the maximum of the CLAMPED objective sits on the kinks the 1e-10 floor puts
        # into it: a homozygous cell of allele frequency f is floored for F < -f / (1 - f), so next to the smooth optimum lie
        # several local maxima ~0.01 apart, and which one a search ends on depends on its path.  The device walks the
        # reference optimiser's own path (Nelder-Mead from the oracle's own start point, nm_advance), so it ends on the oracle's
        # maximum: within 2e-6 (each side stops at a simplex of 1e-6) -- except where two objective values the simplex
        # compares differ by less than their rounding (the oracle adds ~1e5 logs one by one, the device multiplies
        # probabilities and takes one log per segment), which may send the two paths apart once in a few thousand
        # comparisons: at most 0.3 % of the genomes, and those still on a maximum the oracle's own objective rates as good
        # (within 1e-9 of its value).
        close = err <= 2e-6
        at_device = oa.loglikelihood_at(ref, dip, sp, 0, upper, 1, 0.0, 1.0, got["inbred_allele_sum"])
        at_oracle = oa.loglikelihood_at(ref, dip, sp, 0, upper, 1, 0.0, 1.0, freqs[:, 4])
        deficit = at_oracle - at_device
        print(f"Loglikelihood at {G} x {L}: |dF| <= 2e-6 on {int(close.sum())} of {G} genomes, largest {float(err.max()):.3g}; "
              f"largest deficit under the oracle's objective {float(deficit.max()):.3g} log-units = {float((deficit / np.abs(at_oracle)).max()):.3g} of the objective; "
              f"evaluations {kgx.inbreed_last_evaluations()}")
        smooth = f_true[order] >= 0.0
        assert err[smooth].max() <= 2e-6, (float(err[smooth].max()), int(np.flatnonzero(smooth)[err[smooth].argmax()]))
        # measured (round 4, on the moments): every one of the 1000 genomes within 9.6e-7, deficit 6e-14 of the objective
        assert close.sum() >= 0.997 * G, int(close.sum())
        assert (deficit / np.abs(at_oracle))[~close].max(initial=0.0) <= 1e-9, (float(deficit.max()), int(deficit.argmax()))
    m.close()


def test_paired_loglikelihood_search_is_the_single_search(kgx, monkeypatch):
    """Two evaluations per pass (the reflected and the inside-contraction point of a simplex, from one table read) must
    walk the path of the one-at-a-time Nelder-Mead search: the same coefficient for every genome, bit for bit, in fewer
    passes.  Large enough for the multi-kernel table passes and for the compaction of the genomes still searching."""
    G, L = 4096, 40_000                      # 164 M cells: the tail of the search runs on gathered columns (>= 64 M cells left)
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    monkeypatch.setenv("KGX_K7_LL_PASSES", "1")                 # (by default a call of this size runs on the moments)
    paired = m.inbreed(table, "Loglikelihood", phased=True)["inbred_allele_sum"].copy()
    assert kgx.inbreed_last_path() == "loglik passes"
    passes_paired = kgx.inbreed_last_evaluations()
    monkeypatch.setenv("KGX_K7_NO_PAIR", "1")
    single = m.inbreed(table, "Loglikelihood", phased=True)["inbred_allele_sum"].copy()
    passes_single = kgx.inbreed_last_evaluations()
    monkeypatch.delenv("KGX_K7_NO_PAIR")
    assert np.array_equal(paired, single)
    assert passes_paired < passes_single, (passes_paired, passes_single)
    m.close()


def test_window_iteration_kernel_at_its_cell_count_edges(kgx, monkeypatch):
    """k_inbreed_iterate_genome<MODE, CELLS, THREADS> -- a block per genome (2 .. 32 loci per thread) and a wave per
    genome (8 / 16 / 32) -- at the selection sizes where the host switches instantiation, and one past the largest, where
    the multi-kernel passes (HallME: the moments) take over -- against those passes on the same selections of the C5 population: the counts bit
    for bit, HallME to 1e-12 (only the order of the sums differs), Loglikelihood to 2e-6 on genomes with F >= 0 (one
    optimiser, one start, simplex of 1e-6)."""
    G, L = 300, 9000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    f_true = ((np.arange(G) % 101) - 50) / 100.0
    rng = np.random.default_rng(99)
    # (past 2048 loci a wave's lane would hold more than 32 cells: a block per genome whatever the genome count, to 8192)
    for n_sel in (1, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193):
        index = np.sort(rng.choice(L, n_sel, replace=False)).astype(np.uint32)
        sub = np.ascontiguousarray(table[index])
        for algorithm in ("HallME", "Loglikelihood"):
            start = kgx.reference_starts(algorithm, START_SEED, G)
            monkeypatch.setenv("KGX_K7_NO_WAVE", "1")
            monkeypatch.setenv("KGX_K7_HALL_PASSES", "1")       # (HallME: the 50 table passes themselves)
            monkeypatch.setenv("KGX_K7_LL_PASSES", "1")         # (Loglikelihood: the passes, two evaluations each)
            passes = {k: v.copy() for k, v in _fields(m.inbreed(sub, algorithm, phased=True, locus_index=index, start=start)).items()}
            assert kgx.inbreed_last_path() == ("hall passes" if algorithm == "HallME" else "loglik passes")
            monkeypatch.delenv("KGX_K7_NO_WAVE")
            monkeypatch.delenv("KGX_K7_HALL_PASSES")
            monkeypatch.delenv("KGX_K7_LL_PASSES")
            for wave_from in ("1", "1000000"):              # a wave per genome / a block per genome
                monkeypatch.setenv("KGX_K7_WAVE_GENOMES", wave_from)
                fused = _fields(m.inbreed(sub, algorithm, phased=True, locus_index=index, start=start))
                want_path = "one launch" if n_sel <= 8192 else ("hall moments" if algorithm == "HallME" else "loglik moments")
                assert kgx.inbreed_last_path() == want_path, (n_sel, algorithm, kgx.inbreed_last_path())
                monkeypatch.delenv("KGX_K7_WAVE_GENOMES")
                ctx = (n_sel, algorithm, wave_from)
                for name in ("major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"):
                    assert np.array_equal(fused[name], passes[name]), ctx + (name,)
                a, b = fused["inbred_allele_sum"], passes["inbred_allele_sum"]
                both = np.isfinite(a) & np.isfinite(b)
                assert np.array_equal(np.isfinite(a), np.isfinite(b)), ctx
                if algorithm == "HallME":
                    assert np.abs(a[both] - b[both]).max(initial=0.0) <= 1e-12, ctx + (float(np.abs(a[both] - b[both]).max()),)
                elif n_sel >= 255:            # (a handful of loci: a flat objective, several maxima)
                    smooth = both & (f_true >= 0.0)
                    assert np.abs(a[smooth] - b[smooth]).max(initial=0.0) <= 2e-6, ctx + (float(np.abs(a[smooth] - b[smooth]).max()),)
    m.close()


def _fields(result):
    return {name: np.asarray(result[name]) for name in result.dtype.names}


@pytest.mark.gpu
def test_hallme_by_moments_is_the_fifty_passes(kgx, monkeypatch):
    """HallME over a call too large for the one-launch iteration: the per-genome moments of the homozygous cells'
    frequencies (kgx_kernels_hall.h; one pass over the bytes per class of homozygous cell, then 50 steps on the moments)
    against the 50 passes over the bytes (KGX_K7_HALL_PASSES=1) -- phased (1 + amax classes) and unphased (the major
    allele alone), dense and indexed, a genome range off the lane width, both lane widths of the pass: |dF| <= 1e-10
    (the expansion is cut below 1e-12 of a term; 50 steps carry it), everything else bit for bit."""
    G, L = 777, 30_000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    rng = np.random.default_rng(5)
    index = np.sort(rng.choice(L, 17_001, replace=False)).astype(np.uint32)
    for phased in (True, False):
        for sel, g0, g1 in ((None, 0, G), (index, 0, G), (index, 8, 700), (None, 4, 401)):
            sub = table if sel is None else np.ascontiguousarray(table[sel])
            start = kgx.reference_starts("HallME", START_SEED, g1 - g0)
            monkeypatch.setenv("KGX_K7_HALL_PASSES", "1")
            passes = {k: v.copy() for k, v in _fields(m.inbreed(sub, "HallME", phased=phased, locus_index=sel, g0=g0, g1=g1, start=start)).items()}
            monkeypatch.delenv("KGX_K7_HALL_PASSES")
            moments = _fields(m.inbreed(sub, "HallME", phased=phased, locus_index=sel, g0=g0, g1=g1, start=start))
            ctx = (phased, None if sel is None else len(sel), g0, g1)
            for name in passes:
                if name != "inbred_allele_sum":
                    assert np.array_equal(moments[name], passes[name]), ctx + (name,)
            a, b = moments["inbred_allele_sum"], passes["inbred_allele_sum"]
            assert np.array_equal(np.isfinite(a), np.isfinite(b)), ctx
            err = np.abs(a - b)[np.isfinite(b)]
            assert err.max(initial=0.0) <= 1e-10, ctx + (float(err.max()),)
            if phased:
                assert np.abs(b[np.isfinite(b)]).max() > 0.05, ctx      # (not a comparison of zeros; unphased, HallME runs to ~0)
    # A frequency the bins do not reach (0 < y < 2^-20): the call takes the 50 passes, silently and with their result --
    # over more loci than the one-launch iteration holds (8192), so that it IS the moments' path that falls back; what
    # ran is asserted, so that the comparison cannot quietly become one path against itself.  A frequency of 0 (or a
    # negative one: AlleleFreqVector clamps it to 0, _freq.cpp:47) has its own bin: the moments stay.
    n_odd = 9000
    odd_index = np.arange(n_odd, dtype=np.uint32)
    start = kgx.reference_starts("HallME", START_SEED, G)
    for what, value, want_path in (("below the bins", 1e-9, "hall passes"), ("zero", 0.0, "hall moments"), ("negative", -0.25, "hall moments")):
        odd = table[:n_odd].copy()
        odd[7, 0] = value
        odd[8000, 0] = value
        monkeypatch.setenv("KGX_K7_HALL_PASSES", "1")
        want = m.inbreed(odd, "HallME", phased=True, locus_index=odd_index, start=start)["inbred_allele_sum"].copy()
        assert kgx.inbreed_last_path() == "hall passes"
        monkeypatch.delenv("KGX_K7_HALL_PASSES")
        got = m.inbreed(odd, "HallME", phased=True, locus_index=odd_index, start=start)["inbred_allele_sum"].copy()
        assert kgx.inbreed_last_path() == want_path, (what, kgx.inbreed_last_path())
        if want_path == "hall passes":
            assert np.array_equal(got, want), what
        else:
            assert np.abs(got - want).max() <= 1e-10, (what, float(np.abs(got - want).max()))
    m.close()


@pytest.mark.gpu
def test_loglikelihood_by_moments_is_the_passes(kgx, monkeypatch):
    """Loglikelihood over a call too large for the one-launch iteration: the objective from per-genome moments plus the exact
    walk of the cells next to the 1e-10 floor (kgx_kernels_loglik.h) against the table passes over the bytes
    (KGX_K7_LL_PASSES=1, kept as the checker).  (a) EVERY evaluation: the objective at random points -- the whole box,
    both bounds, the kink-ridden negative side -- within 1e-9 of the pass's value, relative; (b) the search: the same
    coefficient, to the bit for almost every genome (one optimiser, one path: a comparison of two objective values that
    differ by less than their rounding may part the two once in thousands), 2e-6 for all; dense and indexed selections, a
    genome range off the workgroups' width; what ran is asserted."""
    G, L = 4096, 40_000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    rng = np.random.default_rng(11)
    index = np.sort(rng.choice(L, 23_001, replace=False)).astype(np.uint32)
    worst = 0.0
    for trial, (sel, g0, g1) in enumerate(((None, 0, G), (index, 0, G), (index, 24, 3001))):
        sub = table if sel is None else np.ascontiguousarray(table[sel])
        n = g1 - g0
        for points in (rng.uniform(-1.0, 1.0, n), rng.uniform(-0.6, 0.05, n), np.where(np.arange(n) % 2 == 0, -1.0, 1.0), np.zeros(n)):
            by_moments = m.inbreed_objective(sub, points, phased=True, locus_index=sel, g0=g0, g1=g1)
            by_passes = m.inbreed_objective(sub, points, phased=True, locus_index=sel, g0=g0, g1=g1, by_passes=True)
            assert np.isfinite(by_moments).all() and np.isfinite(by_passes).all()
            rel = np.abs(by_moments - by_passes) / np.abs(by_passes)
            worst = max(worst, float(rel.max()))
            assert rel.max() <= 1e-9, (trial, float(rel.max()), float(points[rel.argmax()]))
        start = kgx.reference_starts("Loglikelihood", START_SEED, n)
        moments = _fields(m.inbreed(sub, "Loglikelihood", phased=True, locus_index=sel, g0=g0, g1=g1, start=start))
        assert kgx.inbreed_last_path() == "loglik moments", kgx.inbreed_last_path()
        evaluations = kgx.inbreed_last_evaluations()
        monkeypatch.setenv("KGX_K7_LL_PASSES", "1")
        passes = _fields(m.inbreed(sub, "Loglikelihood", phased=True, locus_index=sel, g0=g0, g1=g1, start=start))
        assert kgx.inbreed_last_path() == "loglik passes", kgx.inbreed_last_path()
        monkeypatch.delenv("KGX_K7_LL_PASSES")
        for name in passes:
            if name != "inbred_allele_sum":
                assert np.array_equal(moments[name], passes[name]), (trial, name)
        d = np.abs(moments["inbred_allele_sum"] - passes["inbred_allele_sum"])
        print(f"Loglikelihood by moments, trial {trial}: {int((d == 0).sum())} of {n} genomes to the bit, largest |dF| {float(d.max()):.3g}, "
              f"{evaluations} evaluations at most; objective within {worst:.2e} (relative)")
        assert d.max() <= 2e-6, (trial, float(d.max()), int(d.argmax()))
        assert (d == 0).sum() >= 0.995 * n, (trial, int((d == 0).sum()))
        assert 30 <= evaluations <= 80, evaluations
    # a genome range that does not start on a lane of eight genomes: the passes, said so
    m.inbreed(table, "Loglikelihood", phased=True, g0=4, g1=2000)
    assert kgx.inbreed_last_path() == "loglik passes", kgx.inbreed_last_path()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["KGX_K7_CLASS_BYTES", "KGX_K7_CLASS_SWEEPS"])
def test_class_pass_flavours_agree(kgx, monkeypatch, flavour):
    """The moments' three ways to the same numbers: the shipped one -- ONE pass over the bytes that leaves every class's hits
    as bit rows (k_class_bits), the moments from those on the matrix cores in exact fixed point (k_hall_mfma<., true>) -- against
    a pass over the bytes per class on the matrix cores (KGX_K7_CLASS_BYTES=1) and against the vector sweeps with their fp64 adds
    (KGX_K7_CLASS_SWEEPS=1, round 3's): HallME within 1e-10, Loglikelihood to the bit for almost every genome (the hits' words
    are the same bits, the moments differ in their last places), counts bit for bit.  Genome ranges that end inside a span of
    2048 and inside a lane of eight, that begin off 0, a locus index."""
    G, L = 2600, 30_000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    rng = np.random.default_rng(17)
    index = np.sort(rng.choice(L, 19_003, replace=False)).astype(np.uint32)
    for algorithm, path in (("HallME", "hall moments"), ("Loglikelihood", "loglik moments")):
        for sel, g0, g1 in ((None, 0, G), (index, 8, 2555), (index, 2048, 2600)):
            sub = table if sel is None else np.ascontiguousarray(table[sel])
            start = kgx.reference_starts(algorithm, START_SEED, g1 - g0)
            shipped = {k: v.copy() for k, v in _fields(m.inbreed(sub, algorithm, phased=True, locus_index=sel, g0=g0, g1=g1, start=start)).items()}
            assert kgx.inbreed_last_path() == path, kgx.inbreed_last_path()
            monkeypatch.setenv(flavour, "1")
            other = _fields(m.inbreed(sub, algorithm, phased=True, locus_index=sel, g0=g0, g1=g1, start=start))
            assert kgx.inbreed_last_path() == path, kgx.inbreed_last_path()
            monkeypatch.delenv(flavour)
            ctx = (algorithm, None if sel is None else len(sel), g0, g1)
            for name in shipped:
                if name != "inbred_allele_sum":
                    assert np.array_equal(shipped[name], other[name]), ctx + (name,)
            d = np.abs(shipped["inbred_allele_sum"] - other["inbred_allele_sum"])
            if algorithm == "HallME":
                assert d.max() <= 1e-10, ctx + (float(d.max()),)
            else:
                assert d.max() <= 2e-6 and (d == 0).sum() >= 0.995 * (g1 - g0), ctx + (float(d.max()), int((d == 0).sum()))
    m.close()


@pytest.mark.gpu
def test_moments_with_seven_classes_of_homozygous_cell(kgx, monkeypatch):
    """The moments' pipeline with more classes than the synthetic population has (six alts: the major allele and six alt
    homozygotes, seven bit rows a locus in k_class_bits, seven matrix-core passes): genotypes drawn allele by allele from a
    Dirichlet-like frequency table with every number of alts from one to six, a tenth of the cells with an allele the table has no
    frequency for (index 7) or unknown (15).  HallME within 1e-10 and Loglikelihood within 2e-6 of the passes over the bytes, the
    counts bit for bit; the paths asserted."""
    G, L, amax = 1300, 11_000, 6
    rng = np.random.default_rng(23)
    n_alts = rng.integers(1, amax + 1, L)
    weights = rng.gamma(0.35, 1.0, (L, amax)) * (np.arange(amax)[None, :] < n_alts[:, None])
    major = rng.uniform(0.45, 0.97, L)
    table = weights / weights.sum(axis=1, keepdims=True) * (1.0 - major)[:, None]
    table = np.where(np.arange(amax)[None, :] < n_alts[:, None], np.maximum(table, 2.0e-5), np.nan)      # (inside the bins: >= 2^-20)
    cumulative = np.concatenate([major[:, None], major[:, None] + np.cumsum(np.nan_to_num(table), axis=1)], axis=1)
    cumulative /= cumulative[:, -1:]
    rows = np.zeros((L, G), dtype=np.uint8)
    for phase in range(2):
        draw = rng.random((L, G))
        allele = (draw[:, :, None] >= cumulative[:, None, :]).sum(axis=2).astype(np.uint8)             # 0 = the major allele
        odd = rng.random((L, G))
        allele = np.where(odd < 0.03, 7, np.where(odd < 0.05, 15, allele)).astype(np.uint8)
        rows |= allele << (4 * phase)
    m = kgx.GenotypeMatrix(G, L)
    m.load_rows(rows)
    for algorithm, moments_path, passes_env, passes_path in (("HallME", "hall moments", "KGX_K7_HALL_PASSES", "hall passes"),
                                                              ("Loglikelihood", "loglik moments", "KGX_K7_LL_PASSES", "loglik passes")):
        start = kgx.reference_starts(algorithm, START_SEED, G)
        got = {k: v.copy() for k, v in _fields(m.inbreed(table, algorithm, phased=True, start=start)).items()}
        # (Loglikelihood: heterozygous cells of two alts at 2e-5 each have 2 f1 f2 = 8e-10 -- a search that comes within 1/8 of F = 1
        # meets the floor among them and its genome is handed to the passes)
        assert kgx.inbreed_last_path() in (moments_path, moments_path + " + passes"), kgx.inbreed_last_path()
        monkeypatch.setenv(passes_env, "1")
        want = _fields(m.inbreed(table, algorithm, phased=True, start=start))
        assert kgx.inbreed_last_path() == passes_path, kgx.inbreed_last_path()
        monkeypatch.delenv(passes_env)
        for name in want:
            if name != "inbred_allele_sum":
                assert np.array_equal(got[name], want[name]), (algorithm, name)
        assert int(want["minor_homo_count"].min()) > 0
        d = np.abs(got["inbred_allele_sum"] - want["inbred_allele_sum"])
        assert d.max() <= (1e-10 if algorithm == "HallME" else 2e-6), (algorithm, float(d.max()))
    m.close()


@pytest.mark.gpu
def test_loglikelihood_by_moments_hands_over_what_it_cannot_serve(kgx, monkeypatch):
    """What the statistics cannot give goes to the passes, and comes back with the passes' result: (a) a genome with a
    heterozygous cell whose 2*f1*f2 exceeds 1/2 -- two copies of an allele more frequent than 1/2 on one phase, byte
    (0, a): the upper bound of the clamp can bind -- is searched by passes over its gathered column, the others stay on
    the moments; (b) a frequency without a bin (0 < f < 2^-20) sends the whole call to the passes; (c) an unphased
    population (every alt homozygote such a cell) does not try.  Results: the passes' own, to the bit where the passes ran."""
    G, L = 1500, 12_000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    rows = m.read_rows()
    common = np.flatnonzero(np.nan_to_num(table[:, 0]) > 0.36)[:40]                # loci whose first alt is frequent ...
    table = table.copy()
    table[common, 0] = np.float32(0.58)                                              # ... and now more frequent than 1/2 (the sums stay <= 1)
    table[common, 1:] = np.nan
    rows[common] = np.where((rows[common] & 0xF0) != 0, 0x11, rows[common] & 0x0F)  # (their genotypes: alt 1 alone)
    special = np.array([5, 77, 600, 1499])
    rows[np.ix_(common[:7], special)] = 0x10                                         # (0, 1): both copies on one phase
    m.load_rows(rows)
    start = kgx.reference_starts("Loglikelihood", START_SEED, G)
    monkeypatch.setenv("KGX_K7_LL_PASSES", "1")
    passes = m.inbreed(table, "Loglikelihood", phased=True, start=start)["inbred_allele_sum"].copy()
    monkeypatch.delenv("KGX_K7_LL_PASSES")
    got = m.inbreed(table, "Loglikelihood", phased=True, start=start)["inbred_allele_sum"].copy()
    assert kgx.inbreed_last_path() == "loglik moments + passes", kgx.inbreed_last_path()
    assert np.array_equal(got[special], passes[special])
    d = np.abs(got - passes)
    assert d.max() <= 2e-6 and (d == 0).sum() >= 0.99 * G, (float(d.max()), int((d == 0).sum()))
    objective = m.inbreed_objective(table, np.full(G, -0.2), phased=True)
    assert np.array_equal(np.flatnonzero(np.isnan(objective)), special)
    # (b)
    tiny = table.copy()
    tiny[11, 0] = 1e-9
    tiny[11, 1:] = np.nan
    got = m.inbreed(tiny, "Loglikelihood", phased=True, start=start)["inbred_allele_sum"].copy()
    assert kgx.inbreed_last_path() == "loglik passes", kgx.inbreed_last_path()
    monkeypatch.setenv("KGX_K7_LL_PASSES", "1")
    want = m.inbreed(tiny, "Loglikelihood", phased=True, start=start)["inbred_allele_sum"].copy()
    monkeypatch.delenv("KGX_K7_LL_PASSES")
    assert np.array_equal(got, want)
    # (c)
    m.inbreed(table, "Loglikelihood", phased=False, start=start)
    assert kgx.inbreed_last_path() == "loglik passes", kgx.inbreed_last_path()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["Simple", "RitlandLocus", "HallME", "Loglikelihood"])
def test_batched_windows_are_the_single_calls(kgx, algorithm, monkeypatch):
    """kgx_inbreed_batch -- many (window, super population) tasks in one launch (kgx_kernels_window.h) -- against the same
    tasks as kgx_inbreed calls: the counts bit for bit, the class-frequency sums to 1e-12 (another order of the same
    additions), Simple / RitlandLocus to 1e-10, HallME to 1e-9, Loglikelihood to 2e-6 on genomes with F >= 0 (as between
    any two paths of the library).  Task shapes: genome ranges of every size and offset (a super population's), locus lists
    from 1 to 8192 of a 9000-locus matrix, with and without start points, a dense one (no index); a batch holding a task
    past 8192 loci is made of single calls."""
    G, L = 1310, 9000
    m = kgx.GenotypeMatrix(G, L)
    table = m.synth_multiallelic(1111, 0, 0)
    f_true = ((np.arange(G) % 101) - 50) / 100.0
    rng = np.random.default_rng(17)
    ranges = [(0, 347), (348, 1009), (1012, G), (0, G), (4, 5), (640, 1144)]
    tasks = []
    for k, n_sel in enumerate((1000, 997, 1, 255, 1024, 1025, 2048, 3000, 8192, 1000, 640, 100)):
        g0, g1 = ranges[k % len(ranges)]
        index = np.sort(rng.choice(L, n_sel, replace=False)).astype(np.uint32)
        start = kgx.reference_starts(algorithm, START_SEED + k, g1 - g0) if algorithm in ("HallME", "Loglikelihood") and k % 3 != 2 else None
        tasks.append({"g0": g0, "g1": g1, "locus_index": index, "minor_af": np.ascontiguousarray(table[index]), "start": start})
    tasks.append({"g0": 0, "g1": 600, "locus_index": None, "minor_af": np.ascontiguousarray(table[:1500]), "start": None})
    tolerance = {"Simple": 1e-10, "RitlandLocus": 1e-10, "HallME": 1e-9, "Loglikelihood": 2e-6}[algorithm]

    def compare(batch, label, tasks):
        for k, (t, got) in enumerate(zip(tasks, batch)):
            want = m.inbreed(t["minor_af"], algorithm, phased=True, locus_index=t["locus_index"], g0=t["g0"], g1=t["g1"], start=t["start"])
            ctx = (label, algorithm, k, len(t["minor_af"]), t["g0"], t["g1"])
            for name in ("major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"):
                assert np.array_equal(got[name], want[name]), ctx + (name,)
            for name in ("major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"):
                assert np.allclose(got[name], want[name], rtol=REL, atol=REL), ctx + (name,)
            a, b = got["inbred_allele_sum"], want["inbred_allele_sum"]
            assert np.array_equal(np.isfinite(a), np.isfinite(b)), ctx
            both = np.isfinite(b)
            if algorithm == "Loglikelihood":
                both &= f_true[t["g0"]:t["g1"]] >= 0.0
                if len(t["minor_af"]) < 255:
                    continue                                  # (a handful of loci: a flat objective, several maxima)
            assert np.all(np.abs(a[both] - b[both]) <= tolerance * np.maximum(1.0, np.abs(b[both]))), ctx + (float(np.abs(a[both] - b[both]).max()),)

    small = [t for t in tasks if len(t["minor_af"]) <= 1024]   # (a wave per genome holds at most 1024 loci: KGX_K7_WAVE_LOCI)
    medium = [t for t in tasks if len(t["minor_af"]) <= 2048]
    for subset, wave_loci in ((tasks, None), (small, None), (medium, "2048")):
        for wave_from in (None, "1", "1000000"):              # the library's choice / a wave per genome / a block per genome
            if wave_from:
                monkeypatch.setenv("KGX_K7_WAVE_GENOMES", wave_from)
            if wave_loci:
                monkeypatch.setenv("KGX_K7_WAVE_LOCI", wave_loci)
            batch = m.inbreed_batch(subset, algorithm, phased=True)
            assert kgx.inbreed_last_path() == "one launch"
            monkeypatch.delenv("KGX_K7_WAVE_GENOMES", raising=False)
            monkeypatch.delenv("KGX_K7_WAVE_LOCI", raising=False)
            compare(batch, (len(subset), wave_from, wave_loci), subset)
    # a task the one-launch kernel does not hold: the batch is made of single calls, with their results
    tasks.append({"g0": 0, "g1": 400, "locus_index": None, "minor_af": np.ascontiguousarray(table[:8500]), "start": None})
    batch = m.inbreed_batch(tasks, algorithm, phased=True)
    for t, got in zip(tasks[-2:], batch[-2:]):
        want = m.inbreed(t["minor_af"], algorithm, phased=True, locus_index=t["locus_index"], g0=t["g0"], g1=t["g1"], start=t["start"])
        assert np.array_equal(got["inbred_allele_sum"], want["inbred_allele_sum"], equal_nan=True)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["Simple", "RitlandLocus", "HallME", "Loglikelihood"])
def test_offsets_with_more_than_fourteen_alts_vs_oracle(kgx, algorithm):
    """AlleleFreqVector has no cap on the alts of an offset (kga_analysis_inbreed_freq.cpp:18-57), and isSNP() admits multi-base
    records that differ in one nucleotide (kgl_variant_db.cpp:121-158): three offsets here spell out 20, 17 and 15 SNP alts
    of an eight-base reference.  Their cells go to the matrix's wide rows (8-bit indices, kgx_gt8_set_wide_rows), the
    frequency table gets 20 columns, and every estimator must give what the oracle's generateFrequencies + process* give:
    counts bit for bit, sums to 1e-12, the coefficients as for any other window."""
    G, L = 60, 400
    rec, gt = sv.multiallelic_block(G, L, rng_seed=21, missing_af_frac=0.0, dup_records=0)
    rng = np.random.default_rng(4)
    bases = "ACGT"
    offsets, refs, alts, afs = list(rec.offsets), list(rec.refs), [list(a) for a in rec.alts], [np.asarray(a, dtype=np.float32).reshape(-1, 6) for a in rec.af]
    wide_records = []
    for n_alt, at in ((20, 37), (17, 191), (15, 333)):
        ref = "ACGTTGCA"
        spelled = [ref[:p] + b + ref[p + 1:] for p in range(8) for b in bases if b != ref[p]][:n_alt]
        p = rng.uniform(0.005, 0.04, n_alt)
        refs[at], alts[at], afs[at] = ref, spelled, np.tile(p.astype(np.float32).reshape(-1, 1), (1, 6))
        wide_records.append(at)
    rec = oa.Records(rec.contig, np.array(offsets, dtype=np.uint64), refs, alts, af=afs)
    gt = gt.copy()
    for at in wide_records:                                   # genotypes over all of the offset's alts, both phases
        n_alt = len(alts[at])
        gt[at, :, 0] = np.where(rng.random(G) < 0.5, rng.integers(1, n_alt + 1, G), 0)
        gt[at, :, 1] = np.where(rng.random(G) < 0.5, rng.integers(1, n_alt + 1, G), 0)
        gt[at, :8, 0] = gt[at, :8, 1] = np.arange(13, 21)[:8].clip(max=n_alt)     # homozygotes of alts past the 14th
    ids = sv.genome_ids(G)
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    dip = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    loci = ii.ReferenceLoci(rec)
    amax = max(len(a) for a in loci.alts)
    assert amax == 20
    table = loci.af_table(oa.ALL, amax)
    sel = loci.sample(table, 0, 10**9, 1, 0.0, 1.0)
    bytes_, wide, cells = ii.encode_wide(rec, gt, loci, phased_order=True)
    assert len(wide) == 3 and set(wide.tolist()) <= set(sel.tolist())
    m = kgx.GenotypeMatrix(G, len(loci.offsets))
    m.load_rows(bytes_)
    m.set_wide_rows(wide, cells)
    order = dip.genome_order()
    counts, freqs, present, _ = oa.inbreed_window(ref.filter_snp_pass(), dip, np.full(G, oa.ALL, dtype=np.int32), algorithm, 0, 10**9, 1, 10**6, 0.0, 1.0,
                                                  seed=START_SEED)
    assert present.all()
    got = m.inbreed(table[sel], algorithm, phased=True, locus_index=sel, start=seeded_starts(kgx, algorithm, START_SEED, order))[order]
    for k, name in enumerate(["major_hetero_count", "minor_hetero_count", "minor_homo_count", "major_homo_count", "total_allele_count"]):
        assert np.array_equal(got[name], counts[:, k]), name
    for k, name in enumerate(["major_hetero_freq", "minor_hetero_freq", "minor_homo_freq", "major_homo_freq"]):
        assert np.allclose(got[name], freqs[:, k], rtol=REL, atol=REL), name
    tolerance = {"Simple": 1e-10, "RitlandLocus": 1e-10, "HallME": 1e-9, "Loglikelihood": 2e-6}[algorithm]
    assert np.abs(got["inbred_allele_sum"] - freqs[:, 4]).max() <= tolerance
    # the wide offsets did count: without their rows (bytes 0xFF there: nothing) the totals fall short
    m.set_wide_rows(None, None)
    without = m.inbreed(table[sel], algorithm, phased=True, locus_index=sel, start=seeded_starts(kgx, algorithm, START_SEED, order))[order]
    assert (without["total_allele_count"] < got["total_allele_count"]).any()
    # ... and as a batch of one task (made of single calls: the one-launch kernel reads bytes alone)
    m.set_wide_rows(wide, cells)
    batch = m.inbreed_batch([{"locus_index": sel, "minor_af": table[sel], "start": None if algorithm in ("Simple", "RitlandLocus") else seeded_starts(kgx, algorithm, START_SEED, order)}],
                            algorithm, phased=True)[0][order]
    assert np.array_equal(batch["total_allele_count"], got["total_allele_count"])
    m.close()
