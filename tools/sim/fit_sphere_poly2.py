#!/usr/bin/env python3
"""Coefficients and a numpy fp64 model of the round-2 sphere Gaussian epilogue (csrc/sphere_pairwise.hip, `sphere_gauss_finish`):
   P(z) = asin(sqrt z)^2 / z on z in [0, 1/2]  (phi^2 = 4 z P(z) with z = (1 - |c|) / 2: ONE multiplication after the Horner chain)
   sqrt from the 2^-24 hardware seed by one cubic correction, exp with magic-number rounding and a 256-entry 2^(j/256) table.
Usage: fit_sphere_poly2.py            degree scan (worst relative error of the fp64 Horner evaluation)
       fit_sphere_poly2.py emit DEG   the coefficient line for the kernel
       fit_sphere_poly2.py model DEG  the whole epilogue in numpy against 60-digit arithmetic
Development tool: not imported by the product."""
import sys
import mpmath as mp
import numpy as np
from fit_sphere_poly import cheb_fit, horner64

mp.mp.dps = 60


def p_true(z):
    if z == 0:
        return mp.mpf(1)
    return mp.asin(mp.sqrt(z)) ** 2 / z


def seed_rsq(x, rng):
    """v_rsq_f64 model: 1/sqrt(x) with a relative error up to 2^-24 (tools/ubench.hip)"""
    return (1.0 / np.sqrt(x)) * (1.0 + rng.uniform(-1, 1, x.shape) * 2.0 ** -24)


def sqrt_cubic(x, rng):
    y = seed_rsq(x, rng)
    g = x * y
    r = 1.0 - g * y                       # fma in the kernel
    p = (0.375 * r + 0.5) * r
    return g + g * p


def epilogue_model(c, beta, coef, rng):
    """numpy fp64 model of `sphere_gauss_finish<SCALED>` (no fma): c = inner products, coef = unscaled P coefficients"""
    hi = 1.0 - 1e-15
    zmin = 0.5 * (1.0 - hi)
    z = np.maximum(0.5 - 0.5 * np.abs(c), zmin)
    ws = [cc * (4.0 * beta) for cc in coef]
    w = np.full_like(z, ws[-1])
    for cc in ws[-2::-1]:
        w = w * z + cc
    q = z * w                                              # beta phi^2
    a, b = 2.0 * np.pi * np.sqrt(beta), beta * np.pi ** 2
    t = b - sqrt_cubic(q, rng) * a
    u = np.where(np.signbit(c), 1.0, 0.0)
    y = u * t + q
    L = np.log(2.0) / 256
    Lhi = float(int(L * 2.0 ** 41)) / 2.0 ** 41
    Llo = float(mp.log(2) / 256 - mp.mpf(Lhi))
    magic = 1.5 * 2.0 ** 52
    km = y * (-1.0 / L) + magic
    k = km - magic
    r = (-y - k * Lhi) - k * Llo
    p = r * (1.0 + r * (0.5 + r * (1.0 / 6 + r * (1.0 / 24))))
    ki = (km.view(np.int64) & 0xFFFFFFFF).astype(np.int64)
    ki = np.where(ki >= 2 ** 31, ki - 2 ** 32, ki)
    assert np.array_equal(ki, k.astype(np.int64))
    T = 2.0 ** ((ki & 255) / 256.0)
    return np.ldexp(T + T * p, (ki >> 8).astype(np.int64))


def model(deg, n=400000, beta=1.2931471805599454, seed=0):
    coef = [float(c) for c in cheb_fit(p_true, mp.mpf(0), mp.mpf("0.5"), deg)]
    rng = np.random.default_rng(seed)
    c = np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-16, 0, n // 4), -1 + 10.0 ** rng.uniform(-16, 0, n // 4),
                        np.array([1.0, -1.0, 1 + 2e-16, -1 - 2e-16, 0.0, 0.5, -0.5])])
    lo, hi = -1.0 + 1e-15, 1.0 - 1e-15
    got = epilogue_model(c, beta, coef, rng)
    want = np.exp(-beta * np.arccos(np.clip(c, lo, hi)) ** 2)
    rel = np.abs(got - want) / want
    print(f"deg {deg}: max rel err vs numpy oracle {rel.max():.2e} (at c = {c[rel.argmax()]!r})")
    sub = rng.choice(len(c), 4000, replace=False)
    ex = np.array([float(mp.exp(-mp.mpf(beta) * mp.acos(mp.mpf(float(np.clip(c[i], lo, hi)))) ** 2)) for i in sub])
    print(f"   vs 60-digit reference: model {np.max(np.abs(got[sub] - ex) / ex):.2e}, numpy oracle {np.max(np.abs(want[sub] - ex) / ex):.2e}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "emit":
        coef = cheb_fit(p_true, mp.mpf(0), mp.mpf("0.5"), int(sys.argv[2]))
        print(", ".join(repr(float(c)) for c in coef))
    elif len(sys.argv) > 2 and sys.argv[1] == "model":
        for beta in (1.2931471805599454, 0.05, 40.0):
            model(int(sys.argv[2]), beta=beta)
    else:
        for deg in range(12, 22):
            coef = cheb_fit(p_true, mp.mpf(0), mp.mpf("0.5"), deg)
            zs = np.linspace(0, 0.5, 4001)
            got = horner64(coef, zs)
            want = np.array([float(p_true(mp.mpf(float(z)))) for z in zs])
            print(deg, f"{np.max(np.abs(got - want) / want):.2e}")
