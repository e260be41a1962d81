#!/usr/bin/env python3
"""CPU simulation of the per-lane root-free QL of csrc/spd_eig.hpp at WAVE granularity (64 lanes share the instruction stream):
counts the sweep steps a wave executes for the benchmark distribution (N=4096, d=10) with
  * the round-1 scheme: stage l loops until every lane of the wave has deflated e2[l]; lanes that are done idle;
  * look-ahead shifts: a lane whose e2[l] is already negligible keeps sweeping with the wave (same extent, D-2 .. l) but takes
    its Wilkinson shift from the first block it has NOT deflated yet (l+1 .. l+LA), so the wave's instruction slots that were
    masked off now pre-converge the lane's later stages.
No GPU involved: decides whether the variant is worth building (tools/sim is not part of the product)."""
import sys
import numpy as np

D = 10
EPS2 = 1e-22
ZERO = True
RCPERR = 0.0       # relative error of the reciprocal inside a step (2^-47 = 7e-15 models the one-Newton reciprocal)
PFLOOR = False
FORM2 = False     # gamma = (p a - b gamma_old) / r and p' = f^2 t with t = 1/(p r): two instructions less per step


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    lam = rng.uniform(0.05, 5.0, size=(n, d))
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, lam, q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def tridiag(m):
    """batched Householder tridiagonalisation, column k of the lower triangle eliminated for k = 0..D-3 (spd_eig.hpp order)"""
    a = m.copy()
    n = a.shape[-1]
    for k in range(n - 2):
        x = a[:, k + 1:, k].copy()
        alpha = x[:, 0]
        nrm = np.linalg.norm(x, axis=1)
        u = x.copy()
        u[:, 0] = alpha + np.copysign(nrm, alpha)
        hh = nrm * nrm + np.abs(alpha) * nrm
        inv = np.where(hh == 0, 0.0, 1.0 / np.where(hh == 0, 1.0, hh))
        A22 = a[:, k + 1:, k + 1:]
        p = np.einsum("nij,nj->ni", A22, u) * inv[:, None]
        kap = 0.5 * np.einsum("ni,ni->n", u, p) * inv
        qv = p - kap[:, None] * u
        A22 -= u[:, :, None] * qv[:, None, :] + qv[:, :, None] * u[:, None, :]
        a[:, k + 1, k] = -np.copysign(nrm, alpha)
        a[:, k, k + 1] = a[:, k + 1, k]
        a[:, k + 2:, k] = 0
        a[:, k, k + 2:] = 0
    dg = np.stack([a[:, i, i] for i in range(n)], 1)
    e2 = np.stack([a[:, i + 1, i] ** 2 for i in range(n - 1)] + [np.zeros(a.shape[0])], 1)
    return dg, e2


def nonzero(g):
    return np.copysign(np.maximum(np.abs(g), 1e-75), g)


def ql(dg, e2, lookahead, wave=64):
    """returns (eigenvalues, wave sweep steps total, per-lane-own sweep steps total, sweeps)"""
    dg, e2 = dg.copy(), e2.copy()
    n = dg.shape[0]
    assert n % wave == 0
    flip = np.abs(dg[:, 0]) > np.abs(dg[:, D - 1])
    dg[flip] = dg[flip, ::-1]
    e2[flip, :D - 1] = e2[flip, D - 2::-1]
    steps = 0
    sweeps_w = 0
    lanes = np.arange(n)

    def conv(k):
        return e2[:, k] <= EPS2 * np.abs(dg[:, k] * dg[:, k + 1])

    for l in range(D - 2):
        for it in range(60):
            c_l = conv(l)
            if ZERO:
                e2[c_l, l] = 0.0
            wave_active = (~c_l).reshape(-1, wave).any(1)
            if not wave_active.any():
                break
            steps += int(wave_active.sum()) * (D - 1 - l)
            sweeps_w += int(wave_active.sum())
            lane_in_active_wave = np.repeat(wave_active, wave)
            # which block supplies the shift
            k = np.full(n, l)
            work = ~c_l
            if lookahead:
                ahead = c_l & lane_in_active_wave
                kk = np.full(n, -1)
                for la in range(1, lookahead + 1):
                    if l + la > D - 3:
                        break
                    cv = conv(l + la)
                    if ZERO:
                        e2[ahead & (kk < 0) & cv, l + la] = 0.0
                    cand = ahead & (kk < 0) & ~cv
                    kk[cand] = l + la
                ok = kk >= 0
                k[ok] = kk[ok]
                work = work | ok
            idx = lanes[work]
            kx = k[idx]
            d0, d1, ee = dg[idx, kx], dg[idx, kx + 1], e2[idx, kx]
            delta = 0.5 * (d1 - d0)
            root = np.sqrt(delta * delta + ee)
            sigma = d0 - np.copysign(root - np.abs(delta), delta)
            gamma = (dg[idx, D - 1] - sigma) if PFLOOR else nonzero(dg[idx, D - 1] - sigma)
            p = np.maximum(gamma * gamma, 1e-150) if PFLOOR else gamma * gamma
            c = np.ones(len(idx))
            s = np.zeros(len(idx))
            for i in range(D - 2, l - 1, -1):
                bb = e2[idx, i]
                r = p + bb
                if i != D - 2:
                    e2[idx, i + 1] = s * r
                t = (1.0 / (p * r)) * (1.0 + RCPERR * np.random.default_rng(i).uniform(-1, 1, len(idx)))
                ir = t * p
                c = p * ir
                s = bb * ir
                oldgam = gamma
                al = dg[idx, i]
                if FORM2:
                    f = p * (al - sigma) - bb * oldgam
                    gamma = ir * f
                    dg[idx, i + 1] = oldgam + (al - gamma)
                    p = np.maximum((f * t) * f, 1e-150)
                else:
                    gamma = (c * (al - sigma) - s * oldgam) if PFLOOR else nonzero(c * (al - sigma) - s * oldgam)
                    dg[idx, i + 1] = oldgam + (al - gamma)
                    gr = gamma * r
                    p = np.maximum((gr * t) * gr, 1e-150) if PFLOOR else (gr * t) * gr
            e2[idx, l] = s * p
            dg[idx, l] = sigma + gamma
    a, b2, cc = dg[:, D - 2].copy(), e2[:, D - 2].copy(), dg[:, D - 1].copy()
    sm, df = a + cc, a - cc
    rt = np.sqrt(df * df + 4 * b2)
    r1 = 0.5 * (sm + np.copysign(rt, sm))
    det = a * cc - b2
    dg[:, D - 2] = r1
    dg[:, D - 1] = det / r1
    return dg, steps, sweeps_w


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    x = synth(4096, D, 1234)
    L = np.linalg.cholesky(x)
    Linv = np.linalg.inv(L)
    tot = {}
    for i in range(rows):
        M = np.einsum("ab,nbc,dc->nad", Linv[i], x, Linv[i])
        M = 0.5 * (M + M.transpose(0, 2, 1))
        dg, e2 = tridiag(M)
        ref = np.sort(np.linalg.eigvalsh(M), axis=1)
        for la in (0, 1, 2, 3, 8):
            ev, steps, sweeps = ql(dg, e2, la)
            err = np.max(np.abs(np.sort(ev, 1) - ref) / ref)
            s2 = np.sum(np.log(ev) ** 2, 1)
            r2 = np.sum(np.log(ref) ** 2, 1)
            e2rel = np.max(np.abs(s2 - r2) / np.maximum(r2, 1e-300))
            t = tot.setdefault(la, [0, 0, 0.0, 0.0])
            t[0] += steps
            t[1] += sweeps
            t[2] = max(t[2], err)
            t[3] = max(t[3], e2rel)
    nw = rows * 4096 // 64
    for la, (steps, sweeps, err, e2rel) in tot.items():
        print(f"lookahead {la}: {steps / nw:7.1f} sweep steps / wave, {sweeps / nw:6.2f} sweeps / wave, max rel eig err {err:.2e}, max rel err of sum log^2 {e2rel:.2e}")


if __name__ == "__main__":
    main()
