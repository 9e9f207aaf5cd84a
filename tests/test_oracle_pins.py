"""Pin the oracle (CPU restatement) to what the reference itself guarantees.

The reference ships no tests, golden vectors or fixtures for this path, so the
oracle is "parity unpinned"; these tests hold it to (1) the reference's own conservation identities
(kgl_variant_db_variant.cpp:168-174,221-227,268-275), (2) hand-derived closed forms and the quirks
listed in SURVEY.md §8a, and (3) its synthetic-inbreeding self-check (kga_analysis_inbreed_syngen.cpp).
CPU only.
"""
import numpy as np
import pytest

from . import oracle_api as oa
from . import synth_vcf as sv


def nan6(vals):
    """[n_alt] -> [n_alt][6] with the same AF for every super-population."""
    return np.tile(np.asarray(vals, dtype=np.float32).reshape(-1, 1), (1, 6))


def small_population(mode=oa.Population.PHASED):
    # 3 loci on one contig; offsets chosen so that lexicographic HGVS order != numeric order.
    rec = oa.Records("chr1", [99, 100, 1000], ["A", "C", "G"], [["T"], ["G", "T"], ["A"]],
                     af=[nan6([0.02]), nan6([0.07, np.nan]), nan6([0.6])])
    ids = ["G2", "G0", "G1", "G3"]
    #            locus0   locus1   locus2
    gt = np.array([
        [[1, 0], [0, 0], [1, 1], [0, 0]],      # locus 99:  G2 het, G1 hom
        [[1, 2], [2, 2], [0, 0], [0, 1]],      # locus 100: G2 = alt1/alt2, G0 = alt2/alt2, G3 het alt1
        [[0, 0], [0, 0], [0, 0], [0, 0]],      # locus 1000: nobody carries it
    ], dtype=np.uint8)
    pop = sv.oracle_population(rec, gt, ids, mode)
    return pop, rec, gt, ids


def test_banner_says_unpinned():
    assert b"parity unpinned" in oa.lib().kgo_banner()


def test_variant_index_is_lexicographic_hgvs_and_sparse():
    pop, rec, gt, ids = small_population()
    vdb = oa.VariantDB(pop)
    # Only carried variants exist; hom-ref is implicit; locus 1000 (no carrier) is absent.
    hg = [vdb.hgvs(i) for i in range(vdb.n_variants)]
    assert hg == ["chr1:g.100C>G", "chr1:g.100C>T", "chr1:g.99A>T"]        # "100" sorts before "99"
    assert [vdb.genome_id(i) for i in range(4)] == ["G0", "G1", "G2", "G3"]  # genome id rank
    d = vdb.dosage()
    #              100C>G 100C>T 99A>T
    want = np.array([[0, 2, 0],    # G0
                     [0, 0, 2],    # G1
                     [1, 1, 1],    # G2
                     [1, 0, 0]],   # G3
                    dtype=np.uint8)
    assert np.array_equal(d, want)


def test_conservation_identities():
    pop, *_ = small_population()
    vdb = oa.VariantDB(pop)
    bv = vdb.summary_by_variant()
    bg = vdb.summary_by_genome()
    ps = vdb.population_summary()
    assert np.all(bv.sum(1) == vdb.n_genomes)
    assert np.all(bg.sum(1) == vdb.n_variants)
    assert ps.sum() == vdb.n_genomes * vdb.n_variants
    assert np.array_equal(bv.sum(0), ps) and np.array_equal(bg.sum(0), ps)
    assert vdb.warnings() == 0
    assert np.array_equal(bv, np.array([[2, 2, 0], [2, 1, 1], [2, 1, 1]], dtype=np.uint64))


def test_unphased_parser_counts_the_same_dosage():
    a, *_ = small_population(oa.Population.PHASED)
    b, *_ = small_population(oa.Population.UNPHASED)
    assert np.array_equal(oa.VariantDB(a).dosage(), oa.VariantDB(b).dosage())
    assert a.variant_count() == b.variant_count() == 8


def test_non_diploid_dosage_is_warned_not_counted():
    # Two records at the same offset with the same alt give a genome 3 copies of one HGVS.
    rec = oa.Records("chr1", [10, 10], ["A", "A"], [["T"], ["T"]])
    ids = ["G0", "G1"]
    gt = np.array([[[1, 1], [1, 0]], [[1, 0], [0, 0]]], dtype=np.uint8)
    pop = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    vdb = oa.VariantDB(pop)
    assert np.array_equal(vdb.dosage(), np.array([[3], [1]], dtype=np.uint8))
    bv = vdb.summary_by_variant()
    assert np.array_equal(bv, np.array([[0, 1, 0]], dtype=np.uint64))   # the 3 is dropped
    assert vdb.warnings() >= 1                                            # sum != G


def test_fws_bins_half_open_and_missing_af_in_no_bin():
    pop, rec, gt, ids = small_population()
    variant_out, genome_out, vdb = pop.fws()
    # AF: 99A>T = 0.02 -> bin 0 ; 100C>G = 0.07 -> bin 1 ; 100C>T = missing -> passes both filters, NOT drops it.
    per_bin = genome_out.sum(2)            # [G][11]: every genome sees V_bin variants
    assert np.array_equal(per_bin[0], np.array([1, 1] + [0] * 9, dtype=np.uint64))
    assert np.all(per_bin == per_bin[0])
    # G2 (rank 2) is het for both binned variants; G1 is hom for 99A>T.
    assert np.array_equal(genome_out[2, 0], [0, 1, 0]) and np.array_equal(genome_out[2, 1], [0, 1, 0])
    assert np.array_equal(genome_out[1, 0], [0, 0, 1]) and np.array_equal(genome_out[1, 1], [1, 0, 0])
    assert np.array_equal(variant_out, vdb.summary_by_variant())
    # bin edges: AF == 0.05 belongs to bin 1 ([lo,hi) with >=), AF == 1.0 to no bin.
    rec2 = oa.Records("chr1", [5, 6], ["A", "A"], [["T"], ["T"]], af=[nan6([0.05]), nan6([1.0])])
    gt2 = np.array([[[1, 0]], [[1, 0]]], dtype=np.uint8)
    p2 = sv.oracle_population(rec2, gt2, ["G0"], oa.Population.PHASED)
    _, go, _ = p2.fws()
    assert go[0].sum(1).tolist() == [0, 1] + [0] * 9


def test_hethom_quirks():
    pop, rec, gt, ids = small_population()
    hh = pop.hethom("chr1")     # genome-id order G0..G3; columns: total,snp,indel,hom_minor,het_minor,het_ref_minor,hom_ref
    # G0: locus100 two copies of alt2 -> total 2, unique HGVS 1 -> hom_minor 1, singleton HGVS 0
    assert hh[0].tolist() == [2, 2, 0, 1, 0, 0, 0]
    # G1: locus 99 hom -> same
    assert hh[1].tolist() == [2, 2, 0, 1, 0, 0, 0]
    # G2: locus 99 het (1 variant) -> het_ref_minor 1; locus 100 alt1/alt2 -> "hom_minor" += 2 unique (the quirk), het_minor += 2
    assert hh[2].tolist() == [3, 3, 0, 2, 2, 1, 0]
    assert hh[3].tolist() == [1, 1, 0, 0, 0, 1, 0]
    assert np.all(hh[:, 6] == 0)    # homozygous_reference_alleles_ is never incremented
    # indel classification: REF/ALT of different length
    rec2 = oa.Records("chr1", [5], ["AT"], [["A", "AG", "GT"]])
    gt2 = np.array([[[1, 2], [3, 0]]], dtype=np.uint8)
    p2 = sv.oracle_population(rec2, gt2, ["G0", "G1"], oa.Population.PHASED)
    h2 = p2.hethom("chr1")
    assert h2[0].tolist() == [2, 1, 1, 2, 2, 0, 0]     # "AT>AG" is a SNP (one base differs), "AT>A" an indel
    assert h2[1].tolist() == [1, 1, 0, 0, 0, 1, 0]     # "AT>GT" SNP
    loc = np.array([10, 0, 0, 0, 2, 3, 0], dtype=np.uint64)
    gen = np.array([4, 0, 0, 0, 0, 1, 0], dtype=np.uint64)
    assert oa.lib().kgo_wrights_fis(oa._p(loc), oa._p(gen)) == pytest.approx((0.5 - 0.25) / 0.5, abs=0)


@pytest.mark.parametrize("p,F", [(0.3, 0.0), (0.3, 0.25), (0.01, -0.5), (0.5, 0.5), (0.45, -0.2)])
def test_class_frequencies_closed_form_biallelic(p, F):
    mh, mt, nh, nt = oa.class_frequencies([p], F, normalize=False)
    q = 1.0 - p
    assert nh == F * p + (1.0 - F) * p * p
    assert mh == F * q + (1.0 - F) * q * q
    assert mt == (1.0 - F) * 2.0 * q * p
    assert nt == 0.0
    assert mh + mt + nh + nt == pytest.approx(1.0, abs=1e-15)   # HW+F classes partition unity
    n = oa.class_frequencies([p], F, normalize=True)
    raw = np.maximum(0.0, np.array([mh, mt, nh, nt]))
    assert np.array_equal(n, raw / (raw[0] + raw[1] + raw[2] + raw[3]))


def test_class_frequencies_multiallelic_and_rescale():
    p = [0.1, 0.2, 0.05]
    F = 0.1
    mh, mt, nh, nt = oa.class_frequencies(p, F, normalize=False)
    pm = 1.0 - (0.1 + 0.2 + 0.05)
    assert nh == pytest.approx(sum(F * x + (1 - F) * x * x for x in p), abs=1e-16)
    assert nt == pytest.approx((1 - F) * 2 * (0.1 * 0.2 + 0.1 * 0.05 + 0.2 * 0.05), abs=1e-16)
    assert mh == pytest.approx(F * pm + (1 - F) * pm * pm, abs=1e-16)
    assert mt == pytest.approx((1 - F) * 2 * pm * sum(p), abs=1e-16)
    assert mh + mt + nh + nt == pytest.approx(1.0, abs=1e-15)
    # minor AFs summing over 1 are rescaled by their sum and the major allele vanishes
    mh2, mt2, nh2, nt2 = oa.class_frequencies([0.8, 0.6], 0.0, normalize=False)
    a, b = 0.8 / 1.4, 0.6 / 1.4
    assert (mh2, mt2) == (0.0, 0.0)
    assert nh2 == pytest.approx(a * a + b * b, abs=1e-16) and nt2 == pytest.approx(2 * a * b, abs=1e-16)


def reference_and_diploid():
    """8 genomes x a handful of loci covering every branch of generateFrequencies."""
    #  offset  ref  alts            AF(ALL and all super-pops)
    loci = [
        (100, "A", ["T"],           [0.30]),          # plain biallelic
        (200, "C", ["G", "T"],      [0.10, 0.20]),    # two SNP alts
        (300, "G", ["A", "GT"],     [0.25, 0.05]),    # SNP + insertion (indel never enters INBREED)
        (400, "T", ["C"],           [0.995]),         # major AF 0.005 <= 0.01: hom-ref genomes contribute nothing
        (500, "A", ["C"],           [np.nan]),        # no AF for the alt: invalid locus (empty vector)
        (600, "A", ["G", "C"],      [0.7, 0.6]),      # sum AF > 1 + 1e-5: invalid locus
        (700, "C", ["A", "T"],      [0.15, np.nan]),  # alt2 has no AF: a genome carrying it first is dropped
        (800, "G", ["T"],           [0.40]),
    ]
    rec = oa.Records("chr1", [l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci],
                     af=[nan6(l[3]) for l in loci])
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"], precreate=True)
    ref.add_records(rec, None, oa.Population.REFERENCE)
    ids = sv.genome_ids(8)
    gt = np.zeros((len(loci), 8, 2), dtype=np.uint8)
    gt[0, 0] = (1, 0); gt[0, 1] = (1, 1); gt[0, 2] = (0, 1)
    gt[1, 0] = (1, 2); gt[1, 1] = (2, 2); gt[1, 3] = (2, 0)
    gt[2, 0] = (1, 2); gt[2, 1] = (2, 2); gt[2, 2] = (2, 0); gt[2, 3] = (1, 1)   # alt2 is the indel
    gt[3, 0] = (1, 1); gt[3, 1] = (1, 0)
    gt[4, 0] = (1, 0)
    gt[5, 0] = (1, 2)
    gt[6, 0] = (1, 0); gt[6, 1] = (2, 0); gt[6, 2] = (1, 2); gt[6, 3] = (2, 1)
    gt[7, :] = (1, 1)
    dip = sv.oracle_population(rec, gt, ids, oa.Population.PHASED)
    return ref, dip, rec, gt, ids


def test_generate_frequencies_branches_and_simple_closed_form():
    ref, dip, rec, gt, ids = reference_and_diploid()
    ref_f = ref.filter_snp_pass()
    sp = np.full(8, oa.ALL, dtype=np.int32)
    counts, freqs, present, _ = oa.inbreed_window(ref_f, dip, sp, "Simple", 0, 10_000, 1, 1000, 0.0, 1.0)
    assert present.all()
    # valid loci after SNP filter on the reference: 100, 200, 300 (SNP alt only), 400, 700 (alt1 only), 800.
    # (500 has no AF -> empty vector; 600 sums to 1.3.)   columns: major_het, minor_het, minor_hom, major_hom, total
    # genome 0: 100 het | 200 alt1/alt2 minor-het | 300 SNP alt + indel -> 1 SNP variant: major-het | 400 hom |
    #           700 alt1 het | 800 hom
    assert counts[0].tolist() == [3, 1, 2, 0, 6]
    # genome 1: 100 hom | 200 alt2/alt2 hom | 300 indel/indel -> no SNP variant: major-hom | 400 het |
    #           700 carries only alt2 (no AF, not in the list): front() unmatched -> dropped | 800 hom
    assert counts[1].tolist() == [1, 0, 3, 1, 5]
    # genome 2: 100 het | 200 ref: major-hom | 300 indel only: major-hom | 400 ref but major AF 0.005: dropped |
    #           700 alt1 then alt2(B): two variants, second not in list -> dropped | 800 hom
    assert counts[2].tolist() == [1, 0, 1, 2, 4]
    # genome 3: 100 ref | 200 alt2 het | 300 SNP/SNP hom | 400 dropped | 700 alt2(A) is front(): unmatched, dropped | 800 hom
    assert counts[3].tolist() == [1, 0, 2, 1, 4]
    # genomes 4..7 carry only 800 (hom): 100,200,300,700 major-hom; 400 dropped
    for g in range(4, 8):
        assert counts[g].tolist() == [0, 0, 1, 4, 5]
    # expected class-frequency sums at F=0 over the classified loci, in locus order (genome 4)
    cf = {o: oa.class_frequencies(p, 0.0) for o, p in
          {100: [np.float32(0.30)], 200: [np.float32(0.10), np.float32(0.20)], 300: [np.float32(0.25)],
           700: [np.float32(0.15)], 800: [np.float32(0.40)]}.items()}
    want = np.zeros(4)
    for o in (100, 200, 300, 700, 800):
        want += cf[o]          # majorHom, majorHet, minorHom, minorHet
    got = freqs[4]             # major_het, minor_het, minor_hom, major_hom, F
    assert got[3] == want[0] and got[0] == want[1] and got[2] == want[2] and got[1] == want[3]
    # processSimple: F = (obsHom - expHom) / (N - expHom)
    obs_hom, exp_hom = 1 + 4, want[2] + want[0]
    assert got[4] == (obs_hom - exp_hom) / (5 - exp_hom)


def test_unphased_one_over_one_is_minor_heterozygous():
    ref, dip_phased, rec, gt, ids = reference_and_diploid()
    dip_unphased = sv.oracle_population(rec, gt, ids, oa.Population.UNPHASED)
    ref_f = ref.filter_snp_pass()
    sp = np.full(8, oa.ALL, dtype=np.int32)
    cp, *_ = oa.inbreed_window(ref_f, dip_phased, sp, "Simple", 0, 10_000, 1, 1000, 0.0, 1.0)
    cu, *_ = oa.inbreed_window(ref_f, dip_unphased, sp, "Simple", 0, 10_000, 1, 1000, 0.0, 1.0)
    # homozygous() needs different phases (kgl_variant_db.h:141): UNPHASED 1/1 becomes MINOR_HETEROZYGOUS
    assert np.array_equal(cu[:, 1], cp[:, 1] + cp[:, 2]) and np.all(cu[:, 2] == 0)
    assert np.array_equal(cu[:, [0, 3, 4]], cp[:, [0, 3, 4]])


def test_ritland_closed_form():
    ref, dip, *_ = reference_and_diploid()
    ref_f = ref.filter_snp_pass()
    sp = np.full(8, oa.ALL, dtype=np.int32)
    _, freqs, _, _ = oa.inbreed_window(ref_f, dip, sp, "RitlandLocus", 0, 10_000, 1, 1000, 0.0, 1.0)
    # genome 4: major-hom at 100 (p=.7), 200 (.7), 300 (.75), 700 (.85); minor-hom at 800 (.4)
    f32 = lambda *x: float(sum(np.float32(v).astype(np.float64) for v in x))
    ps = [1.0 - f32(0.30), 1.0 - f32(0.10, 0.20), 1.0 - f32(0.25), 1.0 - f32(0.15), f32(0.40)]
    s = 0.0
    for p in ps:
        s += 1.0 / p
        s -= 1.0
    assert freqs[4, 4] == s / 5


def test_locus_sampling_spacing_and_af_window():
    ref, *_ = reference_and_diploid()
    ref_f = ref.filter_snp_pass()
    all_valid = oa.sample_locii(ref_f, oa.ALL, True, 0, 10**9, 1, 1000, 0.0, 1.0)
    assert all_valid.tolist() == [100, 200, 300, 400, 700, 800]
    # spacing: accept if offset >= prev + spacing (prev = last ACCEPTED), first always accepted
    assert oa.sample_locii(ref_f, oa.ALL, True, 0, 10**9, 250, 1000, 0.0, 1.0).tolist() == [100, 400, 700]
    # count cap and lower bound
    assert oa.sample_locii(ref_f, oa.ALL, True, 150, 10**9, 1, 2, 0.0, 1.0).tolist() == [200, 300]
    # AF window on the summed minor AF; FromTo stops after upper
    assert oa.sample_locii(ref_f, oa.ALL, False, 0, 700, 1, 1000, 0.2, 0.5).tolist() == [100, 200, 300]


def mt19937_64(seed):
    """std::mt19937_64 as the C++ standard defines it ([rand.predef]: w 64, n 312, m 156, r 31, a 0xb5026f5aa96619e9,
    u 29, d 0x5555555555555555, s 17, b 0x71d67fffeda60000, t 37, c 0xfff7eeee00000000, l 43, f 6364136223846793005)."""
    mask = (1 << 64) - 1
    mt = [seed & mask]
    for i in range(1, 312):
        mt.append((6364136223846793005 * (mt[-1] ^ (mt[-1] >> 62)) + i) & mask)
    index = 312
    while True:
        if index == 312:
            for i in range(312):
                x = (mt[i] & 0xFFFFFFFF80000000) | (mt[(i + 1) % 312] & 0x7FFFFFFF)
                mt[i] = mt[(i + 156) % 312] ^ (x >> 1) ^ (0xB5026F5AA96619E9 if x & 1 else 0)
            index = 0
        y = mt[index]
        index += 1
        y ^= (y >> 29) & 0x5555555555555555
        y ^= (y << 17) & 0x71D67FFFEDA60000
        y ^= (y << 37) & 0xFFF7EEEE00000000
        y ^= y >> 43
        yield y & mask


def uniform_real(bits, a, b):
    """libstdc++'s std::uniform_real_distribution<double>(a, b) on one 64-bit draw: generate_canonical<double, 53> =
    double(x) / 2^64 (below 1 by construction), then u * (b - a) + a."""
    u = float(bits) / 18446744073709551616.0
    if u >= 1.0:
        u = float(np.nextafter(1.0, 0.0))
    return u * (b - a) + a


def test_mersenne_twister_known_answer():
    gen = mt19937_64(5489)                       # [rand.predef]: the 10000th invocation of a default-constructed mt19937_64
    for _ in range(9999):
        next(gen)
    assert next(gen) == 9981545732273789042


def test_restart_entropy_is_the_standard_twister_and_the_reference_distributions():
    # processHallME draws UniformRealDistribution(INIT_UPPER_, 0), processLogLikelihood (INIT_UPPER_, INIT_LOWER_)
    # (_calc.cpp:237,166) from a std::mt19937_64, once per restart; task k owns the seed start_seed + k.
    for seed in (11, 20201):
        for algorithm, lower in (("HallME", 0.0), ("Loglikelihood", -0.5)):
            draws = oa.restart_draws(algorithm, seed, 3, restarts=6)
            for k in range(3):
                gen = mt19937_64(seed + k)
                assert draws[k].tolist() == [uniform_real(next(gen), 0.5, lower) for _ in range(6)]


def test_retry_quirk_keeps_the_fifth_restart_of_fifty_em_steps():
    # RetryCalcResult::checkTolerance compares every entry with itself (_calc.cpp:45-68), so the outer loop ends when five
    # restarts are in (MIN_RETRIES_) and the inner one after exactly MINIMUM_ITERATIONS_ = 50 expectation steps: HallME is
    # 50 steps of F <- (1/N) sum_hom F / (F + (1 - F) p) (_calc.cpp:255-285) from the FIFTH draw of the task's twister.
    ref, dip, *_ = reference_and_diploid()
    ref_f = ref.filter_snp_pass()
    sp = np.full(8, oa.ALL, dtype=np.int32)
    seed = 11
    _, fa, _, _ = oa.inbreed_window(ref_f, dip, sp, "HallME", 0, 10_000, 1, 1000, 0.0, 1.0, seed=seed)
    _, fb, _, _ = oa.inbreed_window(ref_f, dip, sp, "HallME", 0, 10_000, 1, 1000, 0.0, 1.0, seed=seed)
    assert np.array_equal(fa, fb)
    # genome 4 (see test_ritland_closed_form): five classified loci, all homozygous, allele frequencies ps
    f32 = lambda *x: float(sum(np.float32(v).astype(np.float64) for v in x))
    ps = [1.0 - f32(0.30), 1.0 - f32(0.10, 0.20), 1.0 - f32(0.25), 1.0 - f32(0.15), f32(0.40)]
    for g in range(4, 8):                                     # genomes 4..7 hold the same genotypes, each its own stream
        gen = mt19937_64(seed + g)
        starts = [uniform_real(next(gen), 0.5, 0.0) for _ in range(5)]
        by_start = []
        for start in starts:
            F = start
            for _ in range(50):
                total = 0.0
                for p in ps:
                    denominator = F + ((1.0 - F) * p)
                    if denominator != 0:
                        total += F / denominator
                F = total / 5.0
            by_start.append(F)
        assert fa[g, 4] == by_start[4], (g, fa[g, 4], by_start)
        assert all(abs(fa[g, 4] - other) > 1e-12 for other in by_start[:4])   # not any of the first four
    assert len({fa[g, 4] for g in range(4, 8)}) == 4           # one stream per task


@pytest.mark.parametrize("algorithm,slope_lo", [("Simple", 0.8), ("Loglikelihood", 0.8), ("HallME", 0.3)])
def test_synthetic_inbreeding_self_check(algorithm, slope_lo):
    # The reference's own validation: genomes generated with known F in [-0.5, 0.5] are estimated back.
    rng = np.random.default_rng(5)
    n = 3000
    offsets = np.arange(1, n + 1, dtype=np.uint64) * 1000
    af = rng.uniform(0.05, 0.5, n).astype(np.float32)
    rec = oa.Records("chr1", offsets, ["A"] * n, [["T"]] * n, af=[nan6([a]) for a in af])
    ref = oa.Population("gnomad")
    ref.add_genomes(["Reference"])
    ref.add_records(rec, None, oa.Population.REFERENCE)
    syn, calc = oa.synthetic_check(ref, oa.ALL, algorithm, 0, 10**9, 1000, 0.0, 1.0, seed=1234)
    assert len(syn) == 101 and syn.min() == -0.5 and syn.max() == 0.5
    slope, intercept = np.polyfit(syn, calc, 1)
    if algorithm == "HallME":
        # EM estimate of IBD sharing is clipped at 0 for outbred genomes; it tracks F for F > 0
        pos = syn > 0.1
        assert np.corrcoef(syn[pos], calc[pos])[0, 1] > 0.9
    else:
        assert slope > slope_lo and abs(intercept) < 0.05
        assert np.abs(calc - syn).max() < 0.12


def test_tuned_cpu_comparator_counts_like_the_port():
    # oracle/kgo_fast.cpp is bench.py's second CPU figure (not the reference's algorithm): it must still count right
    from kgl_gene_amd import capi

    for G in (1, 63, 64, 1003):
        rows, _ = capi.synth_biallelic_host(7, 0, G, 0, 300)
        rows = rows.copy()
        rows[5, 0] |= 0x3            # a non-diploid code
        codes = capi.unpack_dosage2(rows, G)
        want = np.stack([(codes == k).sum(1) for k in range(4)], 1).astype(np.uint32)
        for threads in (1, 3):
            got, seconds, used = oa.fast_count_by_variant(rows, G, threads=threads, repeats=1)
            assert np.array_equal(got, want) and used == threads and seconds > 0


def test_dense_inbreeding_tier_is_generate_frequencies_bit_for_bit():
    """oracle/kgo_inbreed_dense.cpp (the tier the full-size C5 test runs 64 genomes x 5M loci through) against
    generateFrequencies + processSimple / processRitlandLocus on populations both can take: every count, every fp64
    sum and both coefficients are the SAME bits -- same per-locus objects, same summation order."""
    from . import synth_vcf as sv

    for G, L, phased in ((12, 6000, True), (7, 2500, False)):
        d = sv.synth_multiallelic_coded(G, 100, 100 + L, genome_base=37)
        ref = oa.Population("gnomad")
        ref.add_genomes(["Reference"])
        ref.add_records_coded("chr1", d["offsets"], d["ref_code"], d["n_alts"], d["alt_code"], d["af_flat"], None, oa.Population.REFERENCE)
        ref_snp = ref.filter_snp_pass()
        dip = oa.Population("diploid")
        dip.add_genomes(sv.genome_ids(G))
        dip.add_records_coded("chr1", d["offsets"], d["ref_code"], d["n_alts"], d["alt_code"], d["af_flat"], d["alleles"],
                              oa.Population.PHASED if phased else oa.Population.UNPHASED)
        lower, upper = int(d["offsets"][50]), int(d["offsets"][-20])
        for spacing, min_af, max_af in ((1, 0.0, 1.0), (35, 0.05, 0.5)):
            counts, freqs, _ = oa.inbreed_dense(ref, ref_snp, oa.ALL, lower, upper, spacing, min_af, max_af, d["offsets"], d["alleles"], phased=phased)
            for algorithm, column in (("Simple", 4), ("RitlandLocus", 5)):
                c, f, present, _ = oa.inbreed_window(ref_snp, dip, np.full(G, oa.ALL, dtype=np.int32), algorithm, lower, upper, spacing,
                                                     10**9, min_af, max_af)
                assert present.all()
                assert np.array_equal(c, counts)
                assert np.array_equal(f[:, :4], freqs[:, :4])
                assert np.array_equal(f[:, 4], freqs[:, column]), algorithm
            assert counts[:, 4].min() > 10 and counts[:, 1].sum() > 0
            assert (counts[:, 2].sum() > 0) == phased          # unphased 1/1 is not homozygous(): a minor heterozygote


# ---- the optimiser behind Loglikelihood: what the restatement of nlopt's LN_NELDERMEAD assumes ----
# nlopt itself is not in /root/reference (CMakeLists.txt:665 names the library only): these pin the RESTATEMENT's behaviour,
# assumption by assumption, so that a maintainer with nlopt at hand can check each against the real thing.

def test_neldermead_first_simplex_is_a_quarter_of_the_box_turned_inward_at_a_bound():
    # nlopt's default initial step for a bounded variable: (ub - lb) / 4 = 0.5, taken towards the upper bound unless that
    # leaves the box, then towards the lower one
    for x0, second in ((0.0, 0.5), (-0.5, 0.0), (0.5, 1.0), (0.5000001, 0.0000001), (0.9, 0.4), (-1.0, -0.5), (1.0, 0.5)):
        path, _ = oa.neldermead_path(0, 0.123, x0)
        assert path[0] == x0 and abs(path[1] - second) < 1e-15, (x0, path[:2])


def test_neldermead_steps_reflection_expansion_contractions():
    # maximise -(x - 0.3)^2 from 0: simplex {0, 0.5}, values -0.09, -0.04 -> best 0.5, worst 0
    path, result = oa.neldermead_path(0, 0.3, 0.0)
    assert path[:2].tolist() == [0.0, 0.5]
    assert path[2] == 1.0                      # reflection of the worst through the best: 0.5 + (0.5 - 0) = 1.0 (on the bound)
    assert path[3] == 0.25                     # f(1.0) = -0.49 is worse than the worst: inside contraction, halfway best -> worst
    # 0.25 (-0.0025) beats 0.5: the simplex is {0.25 best, 0.5 worst}; reflection 0.0 (-0.09) is worst again: contraction to 0.375
    assert path[4] == 0.0 and path[5] == 0.375
    assert abs(result - 0.3) <= 1e-6
    # expansion: maximise -(x + 0.6)^2 from 0.5: simplex {0.5, 1.0} -> best 0.5; reflection 0.0 is better than the best:
    # the expansion point 0.5 + 2 * (0.5 - 1.0) = -0.5 is tried and, better still, taken; then {-0.5 best, 0.5 worst}: the
    # reflection -1.5 is clamped to -1.0 (between the two): outside contraction halfway from the best to the CLAMPED reflection
    path, result = oa.neldermead_path(0, -0.6, 0.5)
    assert path[:6].tolist() == [0.5, 1.0, 0.0, -0.5, -1.0, -0.75]
    assert abs(result + 0.6) <= 1e-6
    # an expansion that is no better than the reflection is dropped for the reflection: -(x + 0.2)^2 from 0.5
    path, result = oa.neldermead_path(0, -0.2, 0.5)
    assert path[:5].tolist() == [0.5, 1.0, 0.0, -0.5, -0.5]      # reflection 0.0 kept; next simplex {0.0 best, 0.5 worst} reflects to -0.5
    assert abs(result + 0.2) <= 1e-6
    # Pinned on a bound: -(x + 0.9)^2 from 0.5.  After the expansion to -0.5 the clamped reflection -1.0 (-0.01) beats the best
    # (-0.16) and the clamped expansion is the same point; with the best ON the bound the next reflection is clamped onto it,
    # the simplex collapses there, and the search returns -1.0 -- not the maximiser -0.9.  (nlopt's reflectpt() reports a
    # reflected point equal to the centroid and the search stops with XTOL_REACHED at that moment: the same result.)
    path, result = oa.neldermead_path(0, -0.9, 0.5)
    assert path[:5].tolist() == [0.5, 1.0, 0.0, -0.5, -1.0] and np.all(path[4:] == -1.0) and len(path) == 8
    assert result == -1.0
    # outside contraction: -|x - 0.6| from 0.0: simplex {0, 0.5}: best 0.5 (-0.1), worst 0 (-0.6); reflection 1.0 (-0.4) lies
    # between: the outside contraction 0.5 + 0.5 * (1.0 - 0.5) = 0.75 (-0.15) is tried and, not worse than the reflection, taken
    path, _ = oa.neldermead_path(1, 0.6, 0.0)
    assert path[:4].tolist() == [0.0, 0.5, 1.0, 0.75]


def test_neldermead_optimum_on_a_bound_and_the_stopping_rule():
    # a monotone objective: expansions run to the bound; with the best ON the bound the reflected point is clamped onto it, the
    # outside contraction lands there too, and the simplex has collapsed: the search ends on the bound after two more
    # evaluations of the same point (nlopt stops as the clamped reflection equals the centroid: the same result)
    for slope, bound in ((1.0, 1.0), (-1.0, -1.0)):
        for x0 in (-0.5, 0.0, 0.5):
            path, result = oa.neldermead_path(2, slope, x0)
            assert result == bound, (slope, x0, result)
            assert np.all(np.abs(path) <= 1.0)                          # never outside the box
            assert len(path) <= 12 and np.all(path[-2:] == bound), path
    # the stopping rule: the simplex' two points closer than 1e-6 (absolute), never the objective's change
    path, result = oa.neldermead_path(0, 0.3, 0.0)
    assert abs(path[-1] - path[-2]) < 4e-6 and abs(result - 0.3) < 1e-6
    # a flat step objective (all comparisons ties or strict): a tie never replaces the best point
    path, result = oa.neldermead_path(3, 2.0, 0.25)                    # 0 everywhere: every point ties
    assert result == 0.25 and len(path) < 60


def test_loglikelihood_search_from_the_ends_of_the_start_interval_and_with_the_optimum_on_a_bound():
    # processLogLikelihood's starts are drawn from (-0.5, 0.5]: from both ends the search must end on the same coefficient
    # as from the middle on a smooth objective (F >= 0: no cell under the floor); and a genome homozygous everywhere
    # (no heterozygous cell to pull F down) ends ON the upper bound F = 1.
    ref, dip, *_ = reference_and_diploid()
    ref_f = ref.filter_snp_pass()
    sp = np.full(8, oa.ALL, dtype=np.int32)
    results = []
    for seed in (3, 4, 5, 6, 7, 8):
        _, freqs, present, _ = oa.inbreed_window(ref_f, dip, sp, "Loglikelihood", 0, 10_000, 1, 1000, 0.0, 1.0, seed=seed)
        assert present.all()
        results.append(freqs[:, 4].copy())
    results = np.array(results)
    # genome 4 (test_ritland_closed_form): five classified loci, all homozygous -> the likelihood grows with F up to the bound
    assert np.all(results[:, 4] == 1.0), results[:, 4]
    at_one = oa.loglikelihood_at(ref_f, dip, sp, 0, 10_000, 1, 0.0, 1.0, np.full(8, 1.0))
    below = oa.loglikelihood_at(ref_f, dip, sp, 0, 10_000, 1, 0.0, 1.0, np.full(8, 1.0 - 1e-3))
    assert at_one[4] > below[4]
