#!/usr/bin/env python3
"""numpy model of a LOCK-STEP-FRIENDLY dqds on the headline kernel's tridiagonals, beside tools/sim/dqds_lapack_count.py.

Differences to LAPACK's dlasq2, all in dqds' favour for a SIMT kernel:
  * deflation threshold matched to the QL kernel's: e[b-1] <= 1e-20 (q[b] + S) (dlasq2 works to eps^2);
  * the shift is the aggressive one - the smaller eigenvalue mu of the trailing 2x2 of the current L L^T (an UPPER bound of lambda_min by
    interlacing), pulled down by a coupling term: sigma = mu - kappa * e[b-2]-coupling; a sweep that produces a negative d is REPEATED with
    a quarter of the shift and the failed sweep is counted at full cost (in registers a rollback is a select per entry on top of that: not counted);
  * `--oracle`: sigma = lambda_min(active block) (1 - eta) with the TRUE lambda_min (numpy eigvalsh) - what no implementable strategy can beat
    at a given relative shift accuracy eta: eta = 1e-3 is what a Rayleigh-type estimate gives one sweep before convergence.
Counts inner steps per problem (a lane alone) and per wave of 64 (sum over sweeps of the longest active block in the wave: lock step with free
per-lane extents - again in dqds' favour).  Slots: 9 per dqds step, 19 per QL step (tools/ubench_dqds.hip)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ql_lookahead_sim as qs  # noqa: E402
from dqds_lapack_count import chol_qe  # noqa: E402

D = qs.D
TOL = 1e-20


def dqds_sweep(q, e, b, sigma):
    """one dqds transform of the leading block 0..b (inclusive) of every problem; returns new (q, e) and the minimum d"""
    n = q.shape[0]
    qq, ee = q.copy(), e.copy()
    d = q[:, 0] - sigma
    dmin = d.copy()
    for i in range(D - 1):
        act = i < b
        qi = d + e[:, i]
        t = q[:, i + 1] / np.where(qi == 0, 1e-300, qi)
        qq[:, i] = np.where(act, qi, qq[:, i])
        ee[:, i] = np.where(act, e[:, i] * t, ee[:, i])
        d = np.where(act, d * t - sigma, d)
        dmin = np.where(act, np.minimum(dmin, d), dmin)
    idx = np.arange(n)
    qq[idx, b] = d
    return qq, ee, dmin


def trailing_mu(q, e, b):
    """smaller eigenvalue of the trailing 2x2 of T = L L^T restricted to rows b-1, b: [[q_{b-1} + e_{b-2}, sqrt(q_{b-1} e_{b-1})], [., q_b + e_{b-1}]]"""
    idx = np.arange(q.shape[0])
    qb, qa = q[idx, b], q[idx, b - 1]
    eb = e[idx, b - 1]
    ea = np.where(b >= 2, e[idx, np.maximum(b - 2, 0)], 0.0)
    a11, a22, off2 = qa + ea, qb + eb, qa * eb
    tr, det = a11 + a22, a11 * a22 - off2
    disc = np.sqrt(np.maximum((a11 - a22) ** 2 + 4 * off2, 0))
    return det / (0.5 * (tr + disc)), ea


def run(q, e, oracle_eta=None, kappa=1.0):
    n = q.shape[0]
    q, e = q.copy(), e.copy()
    S = np.zeros(n)
    b = np.full(n, D - 1)
    steps = np.zeros(n)
    sweeps = np.zeros(n)
    fails = np.zeros(n)
    wave_steps = 0
    idx = np.arange(n)
    shrink = np.ones(n)
    for it in range(400):
        # deflate
        for _ in range(D):
            conv = (b >= 2) & (e[idx, np.maximum(b - 1, 0)] <= TOL * (q[idx, b] + S))
            if not conv.any():
                break
            q[idx, b] = np.where(conv, q[idx, b] + S, q[idx, b])      # the deflated slot keeps its eigenvalue
            b = np.where(conv, b - 1, b)
        live = b >= 2           # (the last 2x2 is closed form)
        if not live.any():
            break
        bb = np.where(live, b, 2)
        if oracle_eta is not None:
            lam = np.empty(n)
            for k in np.nonzero(live)[0]:
                m = bb[k] + 1
                Lm = np.zeros((m, m))
                Lm[np.arange(m), np.arange(m)] = np.sqrt(q[k, :m])
                Lm[np.arange(1, m), np.arange(m - 1)] = np.sqrt(e[k, :m - 1])
                lam[k] = np.linalg.eigvalsh(Lm @ Lm.T)[0]
            sigma = np.where(live, lam * (1 - oracle_eta), 0.0)
        else:
            mu, ea = trailing_mu(q, e, bb)
            sigma = np.maximum(mu - kappa * ea, 0.0) * shrink
            sigma = np.where(live, sigma, 0.0)
        qq, ee, dmin = dqds_sweep(q, e, bb, sigma)
        ok = dmin > 0
        cost = np.where(live, bb, 0)
        steps += cost
        sweeps += live
        fails += live & ~ok
        wave_steps += int(cost.reshape(-1, 64).max(1).sum())
        take = live & ok
        q = np.where(take[:, None], qq, q)
        e = np.where(take[:, None], ee, e)
        S = np.where(take, S + sigma, S)
        shrink = np.where(live & ~ok, shrink * 0.25, 1.0)
    # closing 2x2 of L L^T: eigenvalues of [[q0, sqrt(q0 e0)], [., q1 + e0]] + S
    a11, a22, off2 = q[:, 0], q[:, 1] + e[:, 0], q[:, 0] * e[:, 0]
    tr, det = a11 + a22, a11 * a22 - off2
    r1 = 0.5 * (tr + np.sqrt((a11 - a22) ** 2 + 4 * off2))
    ev = q.copy()
    ev[:, 0] = r1 + S
    ev[:, 1] = det / r1 + S
    return ev, steps, sweeps, fails, wave_steps


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    x = qs.synth(4096, D, 1234)
    Linv = np.linalg.inv(np.linalg.cholesky(x))
    variants = [("aggressive, kappa=1", dict(kappa=1.0)), ("aggressive, kappa=2", dict(kappa=2.0)), ("aggressive, kappa=0.5", dict(kappa=0.5))]
    if "--oracle" in sys.argv:
        variants += [("oracle eta=1e-3", dict(oracle_eta=1e-3))]      # (eta below ~1e-4 is beyond eigvalsh's ABSOLUTE accuracy once lambda_min - S is tiny)
    for name, kw in variants:
        acc = np.zeros(4)
        werr = 0.0
        nprob = 0
        for i in range(rows):
            M = np.einsum("ab,nbc,dc->nad", Linv[i], x, Linv[i])
            M = 0.5 * (M + M.transpose(0, 2, 1))
            if kw.get("oracle_eta") is not None:
                M = M[:512]
            dg, e2 = qs.tridiag(M)
            q, e = chol_qe(dg, e2)
            ev, steps, sweeps, fails, wave_steps = run(q, e, **kw)
            ref = np.sort(np.linalg.eigvalsh(M), axis=1)
            s2, r2 = np.sum(np.log(ev) ** 2, 1), np.sum(np.log(ref) ** 2, 1)
            werr = max(werr, np.max(np.abs(s2 - r2) / np.maximum(r2, 1.0)))      # (pair (0, 0) is M = I: d^2 = 0)
            acc += np.array([steps.sum(), sweeps.sum(), fails.sum(), wave_steps])
            nprob += len(M)
        st, sw, fl, ws = acc[0] / nprob, acc[1] / nprob, acc[2] / acc[1], acc[3] / (nprob / 64)
        print(f"{name:24s}: lane alone {st:6.1f} steps, {sw:5.1f} sweeps ({100 * fl:4.1f} % failed) -> {9 * st + 12 * sw + 70:5.0f} slots;"
              f" wave of 64 {ws:6.1f} steps -> >= {9 * ws + 12 * sw * ws / st + 70:5.0f} slots; max rel err of sum log^2 {werr:.1e}")
    print("QL today: lane alone 104 steps / 17.0 sweeps -> 2421 slots; wave of 64 with look-ahead 130 steps / 20.5 sweeps -> 3002 slots")


if __name__ == "__main__":
    main()
