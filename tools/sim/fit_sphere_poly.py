#!/usr/bin/env python3
"""Coefficient generation for the sphere Gram epilogue (csrc/sphere_pairwise.hip, Gaussian mode):
   w(z) = asin(sqrt z) / sqrt z  on z in [0, 1/2]   (theta^2 = 4 z w(z)^2 with z = (1 - |c|) / 2)
   exp(r) on |r| <= ln2 / 128 for the table-assisted exponential.
Chebyshev interpolation in 60-digit arithmetic (mpmath), converted to monomials; prints the worst relative error of the fp64
Horner evaluation on a dense grid.  Development tool: not imported by the product."""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    n = deg + 1
    nodes = [mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    xs = [(a + b) / 2 + (b - a) / 2 * t for t in nodes]
    ys = [f(x) for x in xs]
    # solve the Vandermonde system in the scaled variable t, then expand to monomials in x
    A = mp.matrix(n, n)
    for i, t in enumerate(nodes):
        for j in range(n):
            A[i, j] = t ** j
    ct = mp.lu_solve(A, mp.matrix(ys))
    # t = (2x - a - b)/(b - a) = alpha x + beta
    alpha, beta = 2 / (b - a), -(a + b) / (b - a)
    coef = [mp.mpf(0)] * n
    for j in range(n):
        # (alpha x + beta)^j
        for k in range(j + 1):
            coef[k] += ct[j] * mp.binomial(j, k) * alpha ** k * beta ** (j - k)
    return coef


def w_true(z):
    if z == 0:
        return mp.mpf(1)
    s = mp.sqrt(z)
    return mp.asin(s) / s


def horner64(coef, z):
    acc = np.full_like(z, float(coef[-1]))
    for c in coef[-2::-1]:
        acc = acc * z + float(c)
    return acc


if __name__ == "__main__":
    zmax = mp.mpf(sys.argv[1]) if len(sys.argv) > 1 else mp.mpf("0.5")
    for deg in range(10, 24):
        coef = cheb_fit(w_true, mp.mpf(0), zmax, deg)
        zs = np.linspace(0, float(zmax), 4001)
        got = horner64(coef, zs)
        want = np.array([float(w_true(mp.mpf(float(z)))) for z in zs])
        err = np.max(np.abs(got - want) / want)
        print(deg, f"{err:.2e}")


def emit(deg=17):
    """prints the coefficient tables used by csrc/sphere_pairwise.hip"""
    coef = cheb_fit(w_true, mp.mpf(0), mp.mpf("0.5"), deg)
    print("// asin(sqrt z)/sqrt z on [0, 1/2], degree", deg)
    print(", ".join(mp.nstr(c, 20) for c in coef))
    L = mp.log(2) / 64
    # hi part with 33 significant bits: k * hi exact for |k| < 2^20
    hi = mp.mpf(int(L * mp.mpf(2) ** 39)) / mp.mpf(2) ** 39
    print("ln2/64 hi", mp.nstr(hi, 20), "lo", mp.nstr(L - hi, 20), "64/ln2", mp.nstr(1 / L, 20))
    print("table 2^(j/64):")
    print(", ".join(mp.nstr(mp.mpf(2) ** (mp.mpf(j) / 64), 20) for j in range(64)))


def emulate(deg=17, n=200000, beta=1.2931471805599454, seed=0):
    """numpy fp64 emulation of the Gaussian epilogue (no fma) against exp(-beta arccos(clip(c))^2)"""
    coef = [float(c) for c in cheb_fit(w_true, mp.mpf(0), mp.mpf("0.5"), deg)]
    rng = np.random.default_rng(seed)
    c = np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-16, 0, n // 4), -1 + 10.0 ** rng.uniform(-16, 0, n // 4),
                        np.array([1.0, -1.0, 1 + 2e-16, -1 - 2e-16, 0.0, 0.5, -0.5])])
    lo, hi = -1.0 + 1e-15, 1.0 - 1e-15
    zmin = 0.5 * (1.0 - hi)
    z = np.maximum(0.5 - 0.5 * np.abs(c), zmin)
    w = np.full_like(z, coef[-1])
    for cc in coef[-2::-1]:
        w = w * z + cc
    q = (z * w) * w
    s = np.sqrt(q)
    th = np.pi - 2.0 * s
    y = np.where(c < 0, th * th * (-beta), q * (-4.0 * beta))
    y = np.maximum(y, -800.0)
    L = np.log(2.0) / 64
    Lhi = float(int(L * 2.0 ** 39)) / 2.0 ** 39
    Llo = float(mp.log(2) / 64 - mp.mpf(Lhi))
    k = np.rint(y * (1.0 / L))
    r = (y - k * Lhi) - k * Llo
    p = r * (1.0 + r * (0.5 + r * (1.0 / 6 + r * (1.0 / 24 + r * (1.0 / 120)))))
    ki = k.astype(np.int64)
    T = 2.0 ** ((ki & 63) / 64.0)
    got = np.ldexp(T + T * p, (ki >> 6).astype(np.int64))
    want = np.exp(-beta * np.arccos(np.clip(c, lo, hi)) ** 2)
    # exact reference (60 digits) on a subset
    rel = np.abs(got - want) / want
    print(f"deg {deg}: max rel err vs numpy oracle {rel.max():.2e} (at c = {c[rel.argmax()]!r})")
    sub = rng.choice(len(c), 3000, replace=False)
    ex = np.array([float(mp.exp(-mp.mpf(beta) * mp.acos(mp.mpf(float(np.clip(c[i], lo, hi)))) ** 2)) for i in sub])
    print(f"   vs 60-digit reference: emulated {np.max(np.abs(got[sub] - ex) / ex):.2e}, numpy oracle {np.max(np.abs(want[sub] - ex) / ex):.2e}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "emit":
    pass
