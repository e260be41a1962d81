#!/usr/bin/env python3
"""Would binning the problems of a block by difficulty pay for the headline kernel?  CPU simulation on the benchmark
distribution (N = 4096, d = 10) with the wave-granular model of tools/sim/ql_lookahead_sim.py (look-ahead 1 = the kernel):

  * natural order: the 64 lanes of a wave are 64 consecutive columns of one Gram row (the kernel today);
  * ORACLE binning: the problems of a bin (one row = 64 x 64 columns, or a block's 8 rows x 64 columns = 512) sorted by the number of sweep
    steps each would need alone - the best any proxy could do;
  * proxy binning: the same with cheap functions of the tridiagonal (d, e^2) as sort keys.

Prints sweep steps per wave (the QL iteration is ~20 instructions per sweep step of ~4400 per pair-lane; the exchange through LDS that binning
needs is 2 x 20 doubles per problem + the key + the rank computation + the way back of the result).  Development tool."""
import sys
import numpy as np
import ql_lookahead_sim as qs

D = qs.D


def lane_steps(dg, e2):
    """sweep steps each problem needs on its own (wave = 1): per-problem counts"""
    dg, e2 = dg.copy(), e2.copy()
    n = dg.shape[0]
    out = np.zeros(n, dtype=np.int64)
    # run the wave model with wave = 1 on chunks and read the totals per problem through a one-problem-per-wave call would be slow: instead
    # replicate its loop with a per-lane accumulator
    flip = np.abs(dg[:, 0]) > np.abs(dg[:, D - 1])
    dg[flip] = dg[flip, ::-1]
    e2[flip, :D - 1] = e2[flip, D - 2::-1]
    for l in range(D - 2):
        for it in range(60):
            c_l = e2[:, l] <= qs.EPS2 * np.abs(dg[:, l] * dg[:, l + 1])
            e2[c_l, l] = 0.0
            idx = np.nonzero(~c_l)[0]
            if len(idx) == 0:
                break
            out[idx] += D - 1 - l
            d0, d1, ee = dg[idx, l], dg[idx, l + 1], e2[idx, l]
            delta = 0.5 * (d1 - d0)
            root = np.sqrt(delta * delta + ee)
            sigma = d0 - np.copysign(root - np.abs(delta), delta)
            gamma = qs.nonzero(dg[idx, D - 1] - sigma)
            p = gamma * gamma
            c = np.ones(len(idx)); s = np.zeros(len(idx))
            for i in range(D - 2, l - 1, -1):
                bb = e2[idx, i]
                r = p + bb
                if i != D - 2:
                    e2[idx, i + 1] = s * r
                t = 1.0 / (p * r)
                ir = t * p
                c = p * ir
                s = bb * ir
                oldgam = gamma
                al = dg[idx, i]
                gamma = qs.nonzero(c * (al - sigma) - s * oldgam)
                dg[idx, i + 1] = oldgam + (al - gamma)
                gr = gamma * r
                p = (gr * t) * gr
            e2[idx, l] = s * p
            dg[idx, l] = sigma + gamma
    return out


def proxies(dg, e2):
    rel = e2[:, :D - 1] / np.maximum(np.abs(dg[:, :D - 1] * dg[:, 1:]), 1e-300)
    gaps = np.abs(np.diff(np.sort(dg, axis=1), axis=1))
    return {"max e2/|d d'|": rel.max(1), "sum log(1 + e2/|d d'|)": np.log1p(rel).sum(1), "min gap of sorted diagonal": -gaps.min(1),
            "spread of diagonal": dg.max(1) / np.maximum(dg.min(1), 1e-300), "e2[0]/|d0 d1| (first stage)": rel[:, 0]}


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    x = qs.synth(4096, D, 1234)
    Linv = np.linalg.inv(np.linalg.cholesky(x))
    DG, E2 = [], []
    for i in range(rows):
        M = np.einsum("ab,nbc,dc->nad", Linv[i], x, Linv[i])
        dg, e2 = qs.tridiag(0.5 * (M + M.transpose(0, 2, 1)))
        DG.append(dg); E2.append(e2)
    DG, E2 = np.concatenate(DG), np.concatenate(E2)            # row-major: 4096 consecutive problems per Gram row
    n = DG.shape[0]
    own = lane_steps(DG, E2)
    nw = n // 64
    base = qs.ql(DG, E2, 1)[1] / nw
    print(f"{rows} Gram rows = {n} problems.  own steps per problem: mean {own.mean():.1f} (the floor), p50 {np.median(own):.0f}, p90 {np.percentile(own, 90):.0f}, max {own.max()}")
    print(f"natural order, look-ahead 1:                       {base:6.1f} sweep steps per wave")

    def binned(key, bin_size, label):
        order = np.arange(n)
        for b0 in range(0, n, bin_size):
            sl = slice(b0, b0 + bin_size)
            order[sl] = b0 + np.argsort(key[sl], kind="stable")
        st = qs.ql(DG[order], E2[order], 1)[1] / nw
        print(f"{label:50s} {st:6.1f}  ({(st - base) / base * 100:+.1f} % of the QL steps)")
        return st
    # the kernel's natural bins: a wave's 8 rows x 64 columns sit in one block; columns of one row are another candidate (256 = four waves)
    for bs in (128, 256, 512, 4096):
        binned(own.astype(float), bs, f"oracle (own step count), bins of {bs}")
    for name, key in proxies(DG, E2).items():
        for bs in (512,):
            binned(key, bs, f"proxy {name}, bins of {bs}")
    # how much of the own-count variance do the proxies explain?
    for name, key in proxies(DG, E2).items():
        r = np.corrcoef(np.argsort(np.argsort(key)), np.argsort(np.argsort(own)))[0, 1]
        print(f"rank correlation of '{name}' with the own step count: {r:+.2f}")


if __name__ == "__main__":
    main()
