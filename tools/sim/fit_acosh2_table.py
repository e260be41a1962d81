#!/usr/bin/env python3
"""Piecewise table + numpy fp64 model of the round-3 d = 2 affine-invariant Gaussian epilogue (csrc/spd_pairwise_body.hpp, `spd_ai_gauss2_kernel`).

For 2 x 2 SPD matrices M = C C^T (C = W G lower triangular):  log lambda_+- = s +- delta,  s = log sqrt(det M) = log(c00 c11) - a per-POINT
sum -, delta = acosh(tau), tau = tr M / (2 sqrt(det M)) = 1 + u,  u = ((c00 - c11)^2 + c10^2) / (2 c00 c11) >= 0  (no cancellation), so
    d^2 = log^2 lambda_+ + log^2 lambda_- = 2 s^2 + 2 A(tau),   A(tau) = acosh(tau)^2.
A is analytic on all of [1, inf): the square removes acosh's square-root singularity at tau = 1 (A = 2u - u^2/3 + ...); its only singularity
is tau = -1.  So neither a square root nor a logarithm per pair is needed: A comes from a table indexed by the floating-point bits of tau
(binade + M_BITS mantissa bits: slots of relative width 2^-M_BITS), local variable t = tau - (tau with the lower bits cleared), degree DEG.
Usage: fit_acosh2_table.py                  error scan by degree
       fit_acosh2_table.py emit DEG         writes csrc/gabo_acosh2_table.hpp
Development tool: not imported by the product."""
import os
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60
M_BITS = 4           # mantissa bits in the slot index: 16 slots per binade
BINADES = 12         # tau in [1, 2^12): eigenvalue ratio of M up to e^18; beyond it the kernel takes its sqrt + log path (wave-uniform branch)
STRIDE = 10


def slot_bounds(k):
    e, j = divmod(k, 1 << M_BITS)
    lo = mp.mpf(2) ** e * (1 + mp.mpf(j) / (1 << M_BITS))
    return lo, mp.mpf(2) ** e / (1 << M_BITS)


def slot_poly(k, deg):
    """monomial coefficients in t = tau - lo of acosh(tau)^2 on [lo, lo + width]"""
    lo, width = slot_bounds(k)
    n = deg + 1
    nodes = [width / 2 + width / 2 * mp.cos(mp.pi * (2 * i + 1) / (2 * n)) for i in range(n)]
    vals = [mp.acosh(lo + t) ** 2 for t in nodes]
    # solve in the scaled variable t / width (conditioning), then unscale
    A = mp.matrix(n, n)
    for r in range(n):
        for c in range(n):
            A[r, c] = (nodes[r] / width) ** c
    sol = mp.lu_solve(A, mp.matrix(vals))
    return [float(sol[c] / width ** c) for c in range(n)]


def build(deg):
    return np.array([slot_poly(k, deg) for k in range(BINADES << M_BITS)])


def a_model(tau, tab):
    bits = tau.view(np.int64)
    hi = (bits >> 32).astype(np.int64)
    slot = (hi >> (20 - M_BITS)) - (0x3FF << M_BITS)
    lo = ((hi & ~((1 << (20 - M_BITS)) - 1)) << 32).view(np.float64)
    t = tau - lo
    co = tab[slot]
    w = co[:, -1].copy()
    for k in range(tab.shape[1] - 2, -1, -1):
        w = w * t + co[:, k]
    return w


def pairs(n, rng, lo=0.05, hi=5.0, near=False):
    """random 2 x 2 SPD pairs -> (W = chol(A)^-1, G = chol(B)) packed lower"""
    def spd(k):
        th = rng.uniform(0, np.pi, k)
        l1, l2 = rng.uniform(lo, hi, k), rng.uniform(lo, hi, k)
        c, s = np.cos(th), np.sin(th)
        return np.stack([c * c * l1 + s * s * l2, c * s * (l1 - l2), s * s * l1 + c * c * l2], 1)   # a00 a10 a11
    A = spd(n)
    B = spd(n)
    if near:
        B = A * (1 + rng.uniform(-1, 1, A.shape) * 10.0 ** rng.uniform(-15, -3, (n, 1)))
    return A, B


def chol2(S):
    l00 = np.sqrt(S[:, 0])
    l10 = S[:, 1] / l00
    l11 = np.sqrt(S[:, 2] - l10 * l10)
    return l00, l10, l11


def d2_model(A, B, tab):
    a00, a10, a11 = chol2(A)
    w0, w2 = 1 / a00, 1 / a11
    w1 = -a10 * w0 * w2
    g00, g10, g11 = chol2(B)
    c00, c11 = w0 * g00, w2 * g11
    c10 = w1 * g00 + w2 * g10
    inv = (1 / (w0 * w2)) * (1 / (g00 * g11))
    dd = c00 - c11
    u = (dd * dd + c10 * c10) * (0.5 * inv)
    tau = 1.0 + u
    s = np.log(w0 * w2) + np.log(g00 * g11)
    return 2 * s * s + 2 * a_model(tau, tab), tau


def d2_exact(A, B, idx):
    out = []
    for i in idx:
        a = mp.matrix([[A[i, 0], A[i, 1]], [A[i, 1], A[i, 2]]])
        b = mp.matrix([[B[i, 0], B[i, 1]], [B[i, 1], B[i, 2]]])
        L = mp.cholesky(a)
        Li = L ** -1
        m = Li * b * Li.T
        tr, det = m[0, 0] + m[1, 1], m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
        disc = mp.sqrt((m[0, 0] - m[1, 1]) ** 2 + 4 * m[0, 1] * m[1, 0])
        lp, lm = (tr + disc) / 2, det / ((tr + disc) / 2)
        out.append(float(mp.log(lp) ** 2 + mp.log(lm) ** 2))
    return np.array(out)


def scan(degs=(5, 6, 7, 8)):
    rng = np.random.default_rng(0)
    sets = {"bench [0.05, 5]": pairs(200000, rng), "wide [1e-3, 50]": pairs(200000, rng, 1e-3, 50.0), "near-identical": pairs(50000, rng, near=True)}
    for deg in degs:
        tab = build(deg)
        for name, (A, B) in sets.items():
            d2, tau = d2_model(A, B, tab)
            ok = tau < 2.0 ** BINADES
            idx = rng.choice(np.nonzero(ok)[0], 1500, replace=False)
            ex = d2_exact(A, B, idx)
            err = np.abs(d2[idx] - ex)
            print(f"deg {deg} {name}: max abs err of d^2 {err.max():.2e} (d^2 up to {ex.max():.1f}; rel {np.max(err / np.maximum(ex, 1e-300)):.1e}); "
                  f"max abs / max(1, d^2) {np.max(err / np.maximum(1.0, ex)):.2e}; tau max {tau.max():.3g}, in table {ok.mean():.4f}")


def emit(deg):
    tab = build(deg)
    out = [f"// GENERATED by tools/sim/fit_acosh2_table.py emit {deg}: A(tau) = acosh(tau)^2 on [1, 2^{BINADES}) as a piecewise polynomial; slot = the binade and the",
           f"// top {M_BITS} mantissa bits of tau ({1 << M_BITS} slots per binade), local variable t = tau - (tau with the lower mantissa bits cleared), degree {deg},",
           f"// monomial coefficients, {STRIDE} doubles per row (80-byte rows: aligned for ds_read_b128, only rows 16 apart share LDS banks).  A is analytic",
           "// on [1, inf) - the square removes the square-root singularity of acosh at 1 - so no square root and no logarithm is evaluated per pair.",
           "// Error analysis: the same script without arguments.",
           "#pragma once", "namespace gabo {", f"constexpr int kAcosh2Deg = {deg};", f"constexpr int kAcosh2MBits = {M_BITS};",
           f"constexpr int kAcosh2Binades = {BINADES};", "constexpr int kAcosh2Slots = 256;", f"constexpr int kAcosh2Stride = {STRIDE};",
           "// row index = bits 16..23 of tau's high word (the low four exponent bits and the top four mantissa bits): in range by construction, so",
           "// the kernel needs no subtraction of the exponent bias; binade 0 therefore sits in rows 240..255, binades 1..11 in rows 0..175",
           "static __device__ const double kAcosh2Tab[kAcosh2Slots * kAcosh2Stride] = {"]
    rows = np.zeros((256, STRIDE))
    for k in range(BINADES << M_BITS):
        rows[((0x3FF << M_BITS) + k) & 0xFF, :deg + 1] = tab[k]
    for row in rows:
        out.append("    " + ", ".join(float(v).hex() if v != 0 else "0.0" for v in row) + ",")
    out += ["};", "}  // namespace gabo", ""]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gabotorch_amd", "csrc", "gabo_acosh2_table.hpp")
    open(path, "w").write("\n".join(out))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "emit":
        emit(int(sys.argv[2]))
    else:
        scan()
