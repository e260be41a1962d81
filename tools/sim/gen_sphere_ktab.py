#!/usr/bin/env python3
"""Base table + numpy fp64 model of the round-4 sphere Gaussian epilogue (csrc/sphere_pairwise.hip, `sphere_gauss_finish_kt`): the KERNEL VALUE
itself as a piecewise polynomial in v = sqrt((1 + c) / 2) = cos(theta / 2), no exp per output.

  Theta(v) = theta^2 = 4 acos(v)^2 is analytic on [0, 1] (see fit_sphere_piecewise.py).  With S + 1 = 1024 slots centred at s / S, S = 1023, and the local variable
  t = S v - s in [-1/2, 1/2] the block builds, per launch (beta is a launch argument), the Taylor polynomial of
      K(t) = exp(-beta Theta(s / S + t / S)) = exp(p_0) exp(p_1 t + ... + p_6 t^6),   p_k = -beta theta_k(s),
  by the power-series recurrence  e_0 = exp(p_0),  e_m = (1 / m) sum_{k=1..m} k p_k e_{m-k}  (m = 1 .. 6), and folds the t^6 term into the lower
  ones by Chebyshev economisation on [-1/2, 1/2]:  t^6 ~ (768 t^4 - 72 t^2 + 1) / 2048  (error <= 1/2048 instead of 1/64).  The epilogue is then a
  degree-5 Horner chain on three ds_read_b128 - 18 fp64 instructions per output instead of 34.

  This file generates theta_k(s) = Theta^(k)(s / S) / (k! S^k), k = 0 .. 6, in 60-digit arithmetic (recurrence of the ODE
  (1 - v^2) B'' - v B' = 2 for B = acos^2, started from B and B' in closed form; at v = 1 the one-term recurrence) and checks the whole epilogue.

Usage: gen_sphere_ktab.py            model: worst relative error of K against 60-digit arithmetic for several beta
       gen_sphere_ktab.py emit       writes csrc/gabo_sphere_ktab.hpp
Development tool: not imported by the product."""
import os
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
S = 1023          # slots 0 .. S: 1024 rows, one per thread of a 1024-thread block
NK = 7          # theta_0 .. theta_6
QMIN = 4.996003610813204e-16


def taylor_acos2(vs, n):
    """Taylor coefficients b_0..b_n of B(v) = acos(v)^2 at vs (in powers of v - vs)"""
    vs = mp.mpf(vs)
    b = [mp.mpf(0)] * (n + 2)
    if vs == 1:
        # -(k+1)(2k+1) b_{k+1} - k^2 b_k = 2 delta_k0
        b[0] = mp.mpf(0)
        for k in range(0, n):
            b[k + 1] = -(k * k * b[k] + (2 if k == 0 else 0)) / ((k + 1) * (2 * k + 1))
        return b[:n + 1]
    a = mp.acos(vs)
    b[0] = a * a
    b[1] = -2 * a / mp.sqrt(1 - vs * vs)
    for k in range(0, n - 1):
        b[k + 2] = ((2 if k == 0 else 0) + vs * (k + 1) * (2 * k + 1) * b[k + 1] + k * k * b[k]) / ((1 - vs * vs) * (k + 1) * (k + 2))
    return b[:n + 1]


def base_table():
    tab = np.zeros((S + 1, NK))
    for s in range(S + 1):
        b = taylor_acos2(mp.mpf(s) / S, NK - 1)
        for k in range(NK):
            tab[s, k] = float(4 * b[k] / mp.mpf(S) ** k)
    return tab


def build_rows(tab, beta):
    """what the block computes per slot, in fp64 (numpy): the six coefficients of the economised degree-5 polynomial of K"""
    p = -beta * tab                                   # (S+1, 7)
    e = np.zeros_like(p)
    e[:, 0] = np.exp(p[:, 0])
    for m in range(1, NK):
        acc = np.zeros(S + 1)
        for k in range(1, m + 1):
            acc = acc + (k * p[:, k]) * e[:, m - k]
        e[:, m] = acc * (1.0 / m)
    c = e[:, :6].copy()
    c[:, 4] += e[:, 6] * (768.0 / 2048.0)
    c[:, 2] -= e[:, 6] * (72.0 / 2048.0)
    c[:, 0] += e[:, 6] * (1.0 / 2048.0)
    return c


def seed_rsq(x, rng):
    return (1.0 / np.sqrt(x)) * (1.0 + rng.uniform(-1, 1, x.shape) * 2.0 ** -24)


def sqrt_cubic(x, rng):
    y = seed_rsq(x, rng)
    g = x * y
    r = 1.0 - g * y
    p = (0.375 * r + 0.5) * r
    return g + g * p


def epilogue(c, rows, rng):
    q = np.minimum(np.maximum(0.5 * c + 0.5, QMIN), 1.0 - QMIN)
    v = sqrt_cubic(q, rng)
    magic = 6755399441055744.0
    kf = v * S + magic
    kd = kf - magic
    slot = kd.astype(np.int64)
    t = v * S - kd                       # (the kernel: one FMA)
    r = rows[slot]
    w = r[:, 5]
    for k in (4, 3, 2, 1, 0):
        w = w * t + r[:, k]
    return w


def exact(c, beta):
    out = np.empty(len(c))
    for i, ci in enumerate(c):
        cc = min(max(mp.mpf(float(ci)), mp.mpf(-1.0 + 1e-15)), mp.mpf(1.0 - 1e-15))      # the reference's bounds as the doubles they are
        out[i] = float(mp.exp(-mp.mpf(beta) * mp.acos(cc) ** 2))
    return out


def model():
    tab = base_table()
    rng = np.random.default_rng(0)
    c = np.concatenate([rng.uniform(-1, 1, 6000), rng.standard_normal(6000).clip(-3, 3) / 3.2, [1.0, -1.0, 0.0, 1 - 1e-15, -1 + 1e-15, 1 - 1e-9, -1 + 1e-9],
                        1 - 10.0 ** rng.uniform(-14, -1, 500), -1 + 10.0 ** rng.uniform(-14, -1, 500)])
    for beta in (0.2, 0.9, 1.2931471805599454, 2.0, 2.7, 3.5):
        rows = build_rows(tab, beta)
        got = epilogue(c, rows, rng)
        want = exact(c, beta)
        rel = np.abs(got - want) / want
        npy = np.exp(-beta * np.arccos(np.clip(c, -1 + 1e-15, 1 - 1e-15)) ** 2)
        print(f"beta {beta:7.4f}: table epilogue max rel {rel.max():.2e} (at c = {c[rel.argmax()]:+.6f}), numpy acos^2/exp chain {np.max(np.abs(npy - want) / want):.2e}")


def emit():
    tab = base_table()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gabotorch_amd", "csrc", "gabo_sphere_ktab.hpp")
    with open(path, "w") as f:
        f.write("// Generated by tools/sim/gen_sphere_ktab.py emit - do not edit.\n")
        f.write("// theta_k(s) = Theta^(k)(s / S) / (k! S^k), Theta(v) = 4 acos(v)^2, S = %d slots, k = 0 .. %d; row stride 8 doubles (the last one is 0).\n" % (S, NK - 1))
        f.write("#pragma once\nnamespace gabo {\n")
        f.write("constexpr int kSphKtSlots = %d;      // centres s / kSphKtScale, s = 0 .. kSphKtScale\n" % (S + 1))
        f.write("constexpr int kSphKtScale = %d;\n" % S)
        f.write("constexpr int kSphKtBaseStride = 8;\n")
        f.write("__device__ const double kSphKtBase[kSphKtSlots * kSphKtBaseStride] = {\n")
        for s in range(S + 1):
            f.write("    " + ", ".join(repr(float(x)) for x in tab[s]) + ", 0.0,\n")
        f.write("};\n}  // namespace gabo\n")
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "emit":
        emit()
    else:
        model()
