"""The arithmetic behind HallME's moments (kgl_gene_amd/csrc/kgx_kernels_hall.h), restated in numpy and checked against
the direct sum: bins of the frequency axis from the double's exponent and top 7 mantissa bits, moments d^0..d^4 about the
bin's centre, one division and a Horner chain per bin.  No GPU: the device path is compared with the 50 passes and the
oracle in tests/test_inbreed_gpu.py; this pins the expansion's error bound and the bin arithmetic the kernels use."""
import numpy as np

KEY_MANTISSA = 7
MIN_EXPONENT = -20
BINS = 1 + (-MIN_EXPONENT) * (1 << KEY_MANTISSA) + 1
NO_KEY = 0xFFF
MOMENTS = 5


def hall_key(y):
    y = np.asarray(y, dtype=np.float64)
    bits = y.view(np.uint64)
    exponent = ((bits >> np.uint64(52)) & np.uint64(0x7FF)).astype(np.int64) - 1023
    mantissa = ((bits >> np.uint64(52 - KEY_MANTISSA)) & np.uint64((1 << KEY_MANTISSA) - 1)).astype(np.int64)
    key = 1 + (exponent - MIN_EXPONENT) * (1 << KEY_MANTISSA) + mantissa
    bad = ~(y > 0.0) | (exponent < MIN_EXPONENT) | (y > 1.0)
    key = np.where(bad, NO_KEY, key)
    return np.where(y == 0.0, 0, key)


def hall_centre(key):
    key = np.asarray(key, dtype=np.int64)
    k = np.maximum(key - 1, 0)
    exponent = (k >> KEY_MANTISSA) + MIN_EXPONENT + 1023
    bits = (exponent.astype(np.uint64) << np.uint64(52)) | ((k & ((1 << KEY_MANTISSA) - 1)).astype(np.uint64) << np.uint64(52 - KEY_MANTISSA)) \
        | np.uint64(1 << (52 - KEY_MANTISSA - 1))
    return np.where(key == 0, 0.0, bits.view(np.float64))


def test_bins_cover_the_frequency_axis_and_centres_sit_in_the_middle():
    rng = np.random.default_rng(3)
    y = np.concatenate([10.0 ** rng.uniform(-6.0, 0.0, 200_000), [1.0, 2.0 ** MIN_EXPONENT, 0.5, 0.25, np.nextafter(1.0, 0.0)]])
    key = hall_key(y)
    assert key.min() >= 1 and key.max() == BINS - 1 and key.max() < NO_KEY          # y = 1.0 owns the last bin
    c = hall_centre(key)
    assert np.all(np.abs(y - c) <= c * 2.0 ** -(KEY_MANTISSA + 1))                  # relative half-width 2^-8
    inner = key < BINS - 1                                                           # (y = 1.0's bin reaches past 1: its centre is no frequency)
    assert np.all(hall_key(c[inner]) == key[inner])                                  # the centre lies in its own bin
    order = np.argsort(y)
    assert np.all(np.diff(key[order]) >= 0)                                          # bins ascend with y: the sort key is monotone
    assert hall_key(np.array([0.0]))[0] == 0 and hall_centre(np.array([0]))[0] == 0.0
    for bad in (-0.25, 2.0 ** (MIN_EXPONENT - 1), 1.0 + 2.0 ** -40, np.nan, np.inf):
        assert hall_key(np.array([bad]))[0] == NO_KEY, bad                           # the call falls back to the 50 passes


def hall_sum_by_moments(y, F):
    """sum over cells of 1 / (F + (1 - F) * y) from the per-bin moments, as k_hall_iterate evaluates it."""
    key = hall_key(y)
    d = y - hall_centre(key)
    bins = np.unique(key)
    total = 0.0
    u = 1.0 - F
    for b in bins:
        db = d[key == b]
        m = [np.sum(db ** j) for j in range(MOMENTS)]
        q = 1.0 / (u * float(hall_centre(np.array([b]))[0]) + F)
        t = -u * q
        h = m[4]
        for j in (3, 2, 1, 0):
            h = h * t + m[j]
        total += q * h
    return total


def test_expansion_error_is_below_the_bound_for_every_F():
    rng = np.random.default_rng(11)
    # homozygous cells' frequencies the way a genome holds them: mostly the major allele's, some rare alts', a few zeros
    y = np.concatenate([rng.uniform(0.5, 1.0, 40_000), 10.0 ** rng.uniform(-5.5, -0.3, 20_000), np.zeros(50), np.ones(50)])
    bound = (2.0 ** -(KEY_MANTISSA + 1)) ** MOMENTS                                  # 9.1e-13 of a term
    for F in (1e-12, 1e-6, 1e-3, 0.02, 0.25, 0.5, 0.9, 1.0):
        direct = float(np.sum(1.0 / (F + (1.0 - F) * y)))
        got = hall_sum_by_moments(y, F)
        assert abs(got - direct) <= bound * direct + 1e-15 * direct * 64, (F, got, direct)
    # the update itself, 50 times: F <- F * S(F) / N stays within 1e-10 of the direct iteration
    n_total = len(y) * 1.7
    f_direct = f_moments = 0.31
    for _ in range(50):
        f_direct = f_direct * float(np.sum(1.0 / (f_direct + (1.0 - f_direct) * y))) / n_total
        f_moments = f_moments * hall_sum_by_moments(y, f_moments) / n_total
    assert abs(f_direct - f_moments) <= 1e-10 and f_direct > 1e-3, (f_direct, f_moments)


def test_constants_match_the_header():
    import re
    from pathlib import Path

    text = (Path(__file__).resolve().parent.parent / "kgl_gene_amd" / "csrc" / "kgx_kernels_hall.h").read_text()
    assert int(re.search(r"kHallMoments = (\d+);", text).group(1)) == MOMENTS
    assert int(re.search(r"kHallKeyMantissa = (\d+);", text).group(1)) == KEY_MANTISSA
    assert int(re.search(r"kHallMinExponent = (-?\d+);", text).group(1)) == MIN_EXPONENT
    assert int(re.search(r"kHallNoKey = (0x[0-9A-Fa-f]+)u;", text).group(1), 16) == NO_KEY
