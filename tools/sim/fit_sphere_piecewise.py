#!/usr/bin/env python3
"""Piecewise table + numpy fp64 model of the round-3 sphere Gaussian epilogue (csrc/sphere_pairwise.hip, `sphere_gauss_finish_pw`):
   s = sqrt((1 - |c|) / 4) in [1.6e-8, 1/2]   (sin(theta/2) = sqrt(2) s for c >= 0, cos(theta/2) = sqrt(2) s for c < 0)
   theta^2 = Theta_b(s),  b = sign(c):  Theta_+(s) = (2 asin(sqrt2 s))^2,  Theta_-(s) = (pi - 2 asin(sqrt2 s))^2 - both analytic on [0, 1/2]
   per branch 33 slots centred at i/64 (slot = round(64 s)), local variable t = 64 s - slot in [-1/2, 1/2], degree-DEG polynomial per slot.
Usage: fit_sphere_piecewise.py scan          worst relative error of theta^2 by degree
       fit_sphere_piecewise.py model DEG     the whole epilogue in numpy against 60-digit arithmetic
       fit_sphere_piecewise.py emit DEG      writes csrc/gabo_sphere_pw_table.hpp
Development tool: not imported by the product."""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60
NSLOT = 33          # slots 0 .. 32 per branch
SCALE = 64


def theta2(s, neg):
    a = 2 * mp.asin(mp.sqrt(2) * s)
    return (mp.pi - a) ** 2 if neg else a ** 2


def slot_poly(i, neg, deg):
    """monomial coefficients in t of theta^2(i/64 + t/64) on t in [-1/2, 1/2]: Chebyshev interpolation at deg + 1 nodes in 60 digits.
    Slot 0 only ever sees t >= 0 and slot 32 t <= 0 (s in [0, 1/2]): fitted on the half they use."""
    lo, hi = mp.mpf(-0.5), mp.mpf(0.5)
    if i == 0:
        lo = mp.mpf(0)
    if i == NSLOT - 1:
        hi = mp.mpf(0)
    n = deg + 1
    nodes = [(lo + hi) / 2 + (hi - lo) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    vals = [theta2((i + t) / SCALE, neg) for t in nodes]
    # solve the Vandermonde system in 60 digits (n <= 11)
    A = mp.matrix(n, n)
    for r in range(n):
        for c in range(n):
            A[r, c] = nodes[r] ** c
    sol = mp.lu_solve(A, mp.matrix(vals))
    return [sol[k] for k in range(n)]


def build(deg):
    tab = np.zeros((2, NSLOT, deg + 1))
    for b in (0, 1):
        for i in range(NSLOT):
            tab[b, i] = [float(c) for c in slot_poly(i, b == 1, deg)]
    return tab


def seed_rsq(x, rng):
    return (1.0 / np.sqrt(x)) * (1.0 + rng.uniform(-1, 1, x.shape) * 2.0 ** -24)


def sqrt_cubic(x, rng):
    y = seed_rsq(x, rng)
    g = x * y
    r = 1.0 - g * y
    p = (0.375 * r + 0.5) * r
    return g + g * p


def theta2_model(c, tab, rng):
    hi = 1.0 - 1e-15
    qmin = 0.25 * (1.0 - hi)
    q = np.maximum(0.25 - 0.25 * np.abs(c), qmin)
    s = sqrt_cubic(q, rng)
    magic = 1.5 * 2.0 ** 52
    kf = s * SCALE + magic
    kd = kf - magic
    t = s * SCALE - kd
    slot = kd.astype(np.int64)
    b = np.signbit(c).astype(np.int64)
    co = tab[b, slot]
    w = co[:, -1].copy()
    for k in range(tab.shape[2] - 2, -1, -1):
        w = w * t + co[:, k]
    return w


def exp_model(y):
    """exp(-y) as in the kernel: y already scaled would be exact; here the unscaled form of fit_sphere_poly2.py"""
    L = np.log(2.0) / 256
    Lhi = float(int(L * 2.0 ** 41)) / 2.0 ** 41
    Llo = float(mp.log(2) / 256 - mp.mpf(Lhi))
    magic = 1.5 * 2.0 ** 52
    km = y * (-1.0 / L) + magic
    k = km - magic
    r = (-y - k * Lhi) - k * Llo
    p = r * (1.0 + r * (0.5 + r * (1.0 / 6 + r * (1.0 / 24))))
    ki = k.astype(np.int64)
    T = 2.0 ** ((ki & 255) / 256.0)
    return np.ldexp(T + T * p, (ki >> 8).astype(np.int64))


def sample_c(n, rng):
    return np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-16, 0, n // 4), -1 + 10.0 ** rng.uniform(-16, 0, n // 4),
                           np.array([1.0, -1.0, 1 + 2e-16, -1 - 2e-16, 0.0, -0.0, 0.5, -0.5, 1e-300, -1e-300])])


def scan():
    rng = np.random.default_rng(0)
    c = sample_c(200000, rng)
    lo, hi = -1.0 + 1e-15, 1.0 - 1e-15
    sub = rng.choice(len(c), 3000, replace=False)
    ex = np.array([float(mp.acos(mp.mpf(float(np.clip(c[i], lo, hi)))) ** 2) for i in sub])
    for deg in range(6, 11):
        tab = build(deg)
        got = theta2_model(c, tab, rng)
        want = np.arccos(np.clip(c, lo, hi)) ** 2
        print(f"deg {deg}: theta^2 max rel vs numpy {np.max(np.abs(got - want) / want):.2e}; vs 60 digits {np.max(np.abs(got[sub] - ex) / ex):.2e} "
              f"(numpy itself {np.max(np.abs(want[sub] - ex) / ex):.2e})")


def model(deg, n=400000, seed=0):
    tab = build(deg)
    lo, hi = -1.0 + 1e-15, 1.0 - 1e-15
    for beta in (1.2931471805599454, 0.05, 7.2, 40.0):
        rng = np.random.default_rng(seed)
        c = sample_c(n, rng)
        got = exp_model(beta * theta2_model(c, tab, rng))
        want = np.exp(-beta * np.arccos(np.clip(c, lo, hi)) ** 2)
        ok = want > 1e-300
        rel = np.abs(got - want)[ok] / want[ok]
        sub = rng.choice(np.nonzero(ok)[0], 4000, replace=False)
        ex = np.array([float(mp.exp(-mp.mpf(beta) * mp.acos(mp.mpf(float(np.clip(c[i], lo, hi)))) ** 2)) for i in sub])
        print(f"deg {deg} beta {beta}: max rel err vs numpy oracle {rel.max():.2e}; vs 60-digit reference: model "
              f"{np.max(np.abs(got[sub] - ex) / ex):.2e}, numpy oracle {np.max(np.abs(want[sub] - ex) / ex):.2e}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "model":
        model(int(sys.argv[2]))
    elif len(sys.argv) > 2 and sys.argv[1] == "emit":
        import os
        deg = int(sys.argv[2])
        tab = build(deg)
        NEG, STRIDE = 40, 10
        out = ["// GENERATED by tools/sim/fit_sphere_piecewise.py emit %d: theta^2 = acos(c)^2 as a piecewise polynomial of s = sqrt((1 - |c|) / 4) in [0, 1/2]." % deg,
               "// Per branch (c >= 0: theta = 2 asin(sqrt2 s); c < 0: theta = pi - 2 asin(sqrt2 s) - both analytic in s) 33 slots centred at i/64,",
               "// local variable t = 64 s - slot in [-1/2, 1/2], degree %d, monomial coefficients.  Row = slot (c >= 0: slot, c < 0: %d + slot), %d doubles" % (deg, NEG, STRIDE),
               "// per row (coefficients 0..%d, then padding): an 80-byte row stride keeps ds_read_b128 aligned and puts 16 consecutive slots in distinct" % deg,
               "// banks; the offset 40 of the c < 0 rows shifts them by half the bank array, so slot i of one sign never meets slots i-7..i+7 of the other.",
               "// LDS banks.  Model and error analysis (3.8e-15 relative on exp(-beta theta^2) at beta = 1.29 against 60-digit arithmetic; the numpy",
               "// oracle itself 3.4e-15): the same script, `model %d`." % deg,
               "#pragma once", "namespace gabo {", "constexpr int kSphPwDeg = %d;" % deg, "constexpr int kSphPwSlots = 80;",
               "constexpr int kSphPwStride = %d;" % STRIDE, "constexpr int kSphPwNeg = %d;" % NEG,
               "static __device__ const double kSphPwTab[kSphPwSlots * kSphPwStride] = {"]
        for slot in range(80):
            row = np.zeros(STRIDE)
            if slot < NSLOT:
                row[:deg + 1] = tab[0, slot]
            elif NEG <= slot < NEG + NSLOT:
                row[:deg + 1] = tab[1, slot - NEG]
            out.append("    " + ", ".join(float(v).hex() if v != 0 else "0.0" for v in row) + ",")
        out += ["};", "}  // namespace gabo", ""]
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gabotorch_amd", "csrc", "gabo_sphere_pw_table.hpp")
        open(path, "w").write("\n".join(out))
        print("wrote", os.path.normpath(path))
    else:
        scan()
