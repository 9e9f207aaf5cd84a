"""Numerical prototype (numpy, CPU) of the log-likelihood objective evaluated from per-(genome, bin) moments of the
homozygous cells' frequencies plus an exact walk of the cells next to the 1e-10 floor -- the scheme kgx_kernels_loglik.h
implements on the device.  Measures the evaluation error against the direct sum over a grid of F and follows the
reference optimiser's path (1-D Nelder-Mead as the oracle restates it) with both evaluators.

usage: loglik_moments.py [n_loci] [moments J] [log2 of 1/tmax]"""
import math
import sys

import numpy as np

KEY_MANT = 7
MIN_EXP = -20
SMALL = 1e-10


def hall_key(y):
    """bin of y (kgx_kernels_hall.h: hall_key): exponent and top 7 mantissa bits; 0 for y == 0"""
    m, e = np.frexp(y)                       # y = m * 2^e, m in [0.5, 1)
    exponent = e - 1                         # y = 1.m' * 2^exponent
    frac = (m * 2.0 - 1.0)                   # in [0, 1)
    key = 1 + (exponent - MIN_EXP) * (1 << KEY_MANT) + np.floor(frac * (1 << KEY_MANT)).astype(np.int64)
    return np.where(y == 0.0, 0, key)


def hall_centre(key):
    k = key - 1
    exponent = (k >> KEY_MANT) + MIN_EXP
    frac = ((k & ((1 << KEY_MANT) - 1)) + 0.5) / (1 << KEY_MANT)
    return np.where(key == 0, 0.0, np.ldexp(1.0 + frac, exponent))


def direct(F, y_hom, w_het):
    p = F * y_hom + (1.0 - F) * (y_hom * y_hom)
    q = (1.0 - F) * w_het
    return math.fsum(np.log(np.clip(p, SMALL, 1.0))) + math.fsum(np.log(np.clip(q, SMALL, 1.0)))


class Moments:
    def __init__(self, y_hom, w_het, J, tmax):
        self.J, self.tmax = J, tmax
        key = hall_key(y_hom)
        order = np.argsort(key, kind="stable")
        self.key_sorted = key[order]
        self.y_sorted = y_hom[order]
        self.bins = np.unique(self.key_sorted)
        self.begin = np.searchsorted(self.key_sorted, self.bins, side="left")
        self.end = np.searchsorted(self.key_sorted, self.bins, side="right")
        self.c = hall_centre(self.bins)
        self.hw = np.where(self.bins == 0, 0.0, np.ldexp(1.0, ((self.bins - 1) >> KEY_MANT) + MIN_EXP - KEY_MANT - 1))
        d = self.y_sorted - hall_centre(self.key_sorted)
        self.M = np.zeros((len(self.bins), J + 1))
        for b in range(len(self.bins)):
            db = d[self.begin[b]:self.end[b]]
            for j in range(J + 1):
                self.M[b, j] = np.sum(db ** j)
        # sum of log y per bin from the moments (F independent)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.LY = self.M[:, 0] * np.log(np.where(self.c > 0, self.c, 1.0))
            for j in range(1, J + 1):
                self.LY += np.where(self.c > 0, (-1.0) ** (j + 1) * self.M[:, j] / (j * np.where(self.c > 0, self.c, 1.0) ** j), 0.0)
        self.H = len(w_het)
        self.T = math.fsum(np.log(w_het)) if self.H else 0.0
        self.w_min = w_het.min() if self.H else 1.0
        self.exact_cells = 0
        self.evals = 0

    def __call__(self, F):
        self.evals += 1
        u = 1.0 - F
        # the floor: cells with F*y + u*y*y < 1e-10, i.e. y below the positive root
        if u > 0:
            y_floor = (-F + math.sqrt(F * F + 4.0 * u * SMALL)) / (2.0 * u) if F <= 0 else 2.0 * SMALL / (F + math.sqrt(F * F + 4.0 * u * SMALL))
        else:
            y_floor = SMALL / F
        lo, hi = self.c - self.hw, self.c + self.hw
        A = F + u * self.c
        floored = hi <= y_floor * (1.0 - 1e-12)
        with np.errstate(divide="ignore", invalid="ignore"):
            series = (lo > y_floor * (1.0 + 1e-12)) & (u * self.hw <= self.tmax * A)
        total = float(np.sum(self.M[floored, 0])) * math.log(SMALL)
        As = A[series]
        t = u / As
        s = self.M[series, 0] * np.log(As) + self.LY[series]
        for j in range(1, self.J + 1):
            s += (-1.0) ** (j + 1) * self.M[series, j] * t ** j / j
        total += float(np.sum(s))
        for b in np.nonzero(~floored & ~series)[0]:
            yb = self.y_sorted[self.begin[b]:self.end[b]]
            p = F * yb + u * (yb * yb)
            total += float(np.sum(np.log(np.clip(p, SMALL, 1.0))))
            self.exact_cells += len(yb)
        # heterozygous cells: closed form unless some could meet the floor
        if self.H:
            if u * self.w_min >= SMALL:
                total += self.T + self.H * math.log(u)
            elif u * 0.5 <= SMALL:
                total += self.H * math.log(SMALL)
            else:
                raise RuntimeError("het floor between: fallback")
        return total


def neldermead(f, x0):
    n = 0
    clampx = lambda x: min(1.0, max(-1.0, x))
    step = 0.5
    xa = clampx(x0)
    xb = xa + step
    if xb > 1.0:
        xb = xa - step
    xb = clampx(xb)
    path = [xa, xb]
    fa, fb = f(xa), f(xb)
    n = 2
    while n < 500:
        if fb > fa:
            xa, xb, fa, fb = xb, xa, fb, fa
        if abs(xa - xb) < 1e-6:
            break
        xr = clampx(xa + (xa - xb)); fr = f(xr); n += 1; path.append(xr)
        if fr > fa:
            xe = clampx(xa + 2.0 * (xa - xb)); fe = f(xe); n += 1; path.append(xe)
            if fe > fr: xb, fb = xe, fe
            else: xb, fb = xr, fr
        elif fr > fb:
            xc = clampx(xa + 0.5 * (xr - xa)); fc = f(xc); n += 1; path.append(xc)
            if fc >= fr: xb, fb = xc, fc
            else: xb, fb = xr, fr
        else:
            xc = xa + 0.5 * (xb - xa); fc = f(xc); n += 1; path.append(xc)
            xb, fb = xc, fc
    return (xa if fa >= fb else xb), path


def genome(rng, n_loci, F_true):
    """C5-like: 1..3 alts per locus, AF U[0.01, 0.5] rescaled to sum <= 0.6; genotype classes by the reference's model"""
    n_alt = rng.choice([1, 2, 3], size=n_loci, p=[0.7, 0.2, 0.1])
    y_hom, w_het = [], []
    af = rng.uniform(0.01, 0.5, size=(n_loci, 3)).astype(np.float32).astype(np.float64)
    af[np.arange(3)[None, :] >= n_alt[:, None]] = 0.0
    s = af.sum(axis=1)
    af *= np.where(s > 0.6, 0.6 / s, 1.0)[:, None]
    pm = 1.0 - af.sum(axis=1)
    r = rng.random(n_loci)
    alleles = np.concatenate([pm[:, None], af], axis=1)        # allele 0 = major
    # draw two alleles with inbreeding: with prob F identical by descent
    ibd = rng.random(n_loci) < max(F_true, 0.0)
    cdf = np.cumsum(alleles, axis=1)
    a1 = (rng.random(n_loci)[:, None] > cdf).sum(axis=1).clip(0, 3)
    a2 = np.where(ibd, a1, (rng.random(n_loci)[:, None] > cdf).sum(axis=1).clip(0, 3))
    if F_true < 0:                                             # fewer homozygotes: turn a share of them into heterozygotes
        flip = (a1 == a2) & (rng.random(n_loci) < -F_true)
        a2 = np.where(flip, (a1 + 1) % (n_alt + 1), a2)
    f1 = alleles[np.arange(n_loci), a1]
    f2 = alleles[np.arange(n_loci), a2]
    hom = a1 == a2
    keep_hom = hom & ((a1 != 0) | (pm > 0.01)) & (f1 > 0)
    het = ~hom & (f1 > 0) & (f2 > 0)
    return f1[keep_hom], 2.0 * f1[het] * f2[het]


def main():
    n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    J = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tmax = 2.0 ** -float(sys.argv[3]) if len(sys.argv) > 3 else 2.0 ** -6
    rng = np.random.default_rng(5)
    worst_rel, worst_abs = 0.0, 0.0
    for F_true in (-0.45, -0.3, -0.1, -0.02, 0.0, 0.05, 0.3):
        y_hom, w_het = genome(rng, n_loci, F_true)
        mom = Moments(y_hom, w_het, J, tmax)
        err_abs, err_rel = 0.0, 0.0
        grid = np.concatenate([np.linspace(-1, 1, 81), rng.uniform(-0.6, 0.2, 60)])
        for F in grid:
            a, b = direct(F, y_hom, w_het), mom(F)
            err_abs = max(err_abs, abs(a - b)); err_rel = max(err_rel, abs(a - b) / abs(a))
        cells_per_eval = mom.exact_cells / mom.evals
        mom.exact_cells = mom.evals = 0
        dF, same = 0.0, 0
        starts = rng.uniform(-0.5, 0.5, 12)
        for x0 in starts:
            xd, pd = neldermead(lambda F: direct(F, y_hom, w_het), x0)
            xm, pm = neldermead(mom, x0)
            dF = max(dF, abs(xd - xm)); same += int(pd == pm)
        print(f"F_true {F_true:+.2f}: {len(y_hom)} hom, {len(w_het)} het cells, {len(mom.bins)} bins; grid: |err| {err_abs:.2e} abs {err_rel:.2e} rel, "
              f"{cells_per_eval:.0f} exact cells/eval; search: |dF| max {dF:.2e}, {same}/{len(starts)} identical paths, "
              f"{mom.exact_cells / max(mom.evals, 1):.0f} exact cells/eval, {mom.evals / len(starts):.1f} evals", flush=True)
        worst_rel, worst_abs = max(worst_rel, err_rel), max(worst_abs, err_abs)
    print(f"J = {J}, tmax = 2^{math.log2(tmax):.1f}: worst {worst_abs:.2e} abs, {worst_rel:.2e} rel")


main()
