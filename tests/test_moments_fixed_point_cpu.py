"""The arithmetic the matrix-core moment pass rests on (kgx_kernels_hall.h: k_hall_digits, k_hall_mfma), restated in exact integers:
a power t^j of a slot's scaled distance from its bin centre, t in [-1/2, 1/2), is round(t^j * 2^54) written in seven balanced
base-256 digits; a product with 0 / -128 / 2^q hits summed over an item's slots is then an exact int32 per digit column, and the digit
sums put together again are the sum of the fixed-point powers -- off the real sum by at most 2^-55 a slot.  (The kernels themselves are
checked on the GPU against the passes over the bytes: tests/test_inbreed_gpu.py.)"""
from fractions import Fraction

import numpy as np

DIGITS, BITS, ITEM_SLOTS = 7, 54, 2048          # kHallDigits, kHallDigitBits, kHallItemLoci


def balanced_digits(v: int) -> list[int]:
    out = []
    for _ in range(DIGITS):
        low = ((v + 128) & 255) - 128
        out.append(low)
        v = (v - low) >> 8
    assert v == 0, "seven digits hold every |V| <= 2^53"
    return out


def test_balanced_digits_hold_the_fixed_point_powers_exactly():
    rng = np.random.default_rng(3)
    ts = np.concatenate([rng.uniform(-0.5, 0.5, 2000), [-0.5, 0.0, np.nextafter(0.5, 0.0), 2.0**-30, -(2.0**-52)]])
    for t in ts:
        for j in (1, 2, 3, 4):
            power = float(t) ** j if j == 1 else float(np.prod(np.full(j, t)))      # (any rounding of the double product: the kernel's is one of them)
            v = int(np.rint(power * 2.0**BITS))
            assert abs(v) <= 2 ** (BITS - j)                                         # |t^j| <= 2^-j: the top digit stays small
            digits = balanced_digits(v)
            assert all(-128 <= d <= 127 for d in digits)
            assert sum(d << (8 * k) for k, d in enumerate(digits)) == v
            assert abs(Fraction(v, 2**BITS) - Fraction(power)) <= Fraction(1, 2 ** (BITS + 1))


def test_an_items_digit_sums_fit_int32_and_give_back_the_sum():
    rng = np.random.default_rng(4)
    t = rng.uniform(-0.5, 0.5, ITEM_SLOTS)
    hit = rng.random(ITEM_SLOTS) < 0.3
    for weight in (-128, 1, 64):                                                     # a hit from the bytes; from the bit rows: 2^q, q = 7 as -128
        for j in (1, 2, 3, 4):
            values = [int(np.rint(float(x) ** j * 2.0**BITS)) for x in t]
            columns = np.array([balanced_digits(v) for v in values], dtype=np.int64)         # [slot][digit]
            sums = (columns * (hit[:, None] * weight)).sum(axis=0)                            # what the MFMA accumulates, per digit column
            assert np.abs(sums).max() < 2**31
            assert np.abs(columns).max() * abs(weight) * ITEM_SLOTS < 2**31                   # ... and whatever the hits: no overflow
            together = sum(int(s) << (8 * k) for k, s in enumerate(sums))
            assert together == weight * sum(v for v, h in zip(values, hit) if h)
