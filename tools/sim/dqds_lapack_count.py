#!/usr/bin/env python3
"""How many inner-loop steps does a PRODUCTION dqds need on the tridiagonals of the headline kernel?  

LAPACK's dlasq2 (the dqds behind dbdsqr / dlasq1: Parlett-Marques shifts `dlasq4`, failure handling, flipping, early 2x2 deflation) is
called through ctypes on the Cholesky factors (q, e) of the same tridiagonal matrices the QL simulation uses (tools/sim/ql_lookahead_sim.py:
benchmark distribution, N = 4096, d = 10).  dlasq2 returns its own counters in Z(2N+3..2N+5): sweeps, divisions (= inner-loop steps, one
division per step) / N^2, percentage of failed shifts.  That is the step count of the best serial shift strategy there is, per problem, with
NO lock-step tax - a lower bound for anything a wave of 64 lanes could do with it.

The comparison is in ISSUE SLOTS: one dqds step is 9 slots on gfx950 (tools/ubench_dqds.hip: add, v_rcp_f64 at 3 slots, 2 refinement FMAs, 3 products /
FMAs), one root-free QL step 19 (16 VALU + the reciprocal's 3).  One QL sweep is algebraically TWO Cholesky-LR (= dqds) sweeps with the same
shift, so the halved step cost does not by itself buy anything; what is measured here is whether fresher shifts do.

No GPU involved; tools/sim is not part of the product."""
import ctypes
import glob
import os
import sys

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ql_lookahead_sim as qs  # noqa: E402

D = qs.D


def _lapack():
    libs = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas-*.so"))
    if not libs:
        raise SystemExit("no bundled openblas with LAPACK found")
    lib = ctypes.CDLL(libs[0])
    f = lib.scipy_dlasq2_
    f.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    f.restype = None
    return f


def chol_qe(dg, e2):
    """T = L L^T, L lower bidiagonal: q_i = l_ii^2, e_i = l_{i+1,i}^2 (T SPD): q_0 = a_0, e_i = b_i^2 / q_i, q_{i+1} = a_{i+1} - e_i"""
    n = dg.shape[1]
    q = np.empty_like(dg)
    e = np.zeros_like(dg)
    q[:, 0] = dg[:, 0]
    for i in range(n - 1):
        e[:, i] = e2[:, i] / q[:, i]
        q[:, i + 1] = dg[:, i + 1] - e[:, i]
    return q, e


def dlasq2_counts(q, e, f):
    n = q.shape[1]
    out = np.empty((q.shape[0], 3))
    ev = np.empty_like(q)
    for k in range(q.shape[0]):
        z = np.zeros(4 * n)
        z[0:2 * n:2] = q[k]
        z[1:2 * n - 1:2] = e[k, :n - 1]
        nn = ctypes.c_int(n)
        info = ctypes.c_int(0)
        f(ctypes.byref(nn), z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(info))
        assert info.value == 0, info.value
        ev[k] = z[:n]
        # (dlasq3 adds N0 - I0 + 2 to NDIV per sweep - two more than the steps of the sweep - and dlasq2 starts at 2 (N0 - I0): taken off here)
        it = z[2 * n + 2]
        out[k] = (it, z[2 * n + 3] * n * n - 2.0 * it, z[2 * n + 4])
    return ev, out


def ql_lone_lane_steps(dg, e2):
    """the kernel's QL (FORM2, p floor) with every lane alone: wave = 1"""
    qs.FORM2 = True
    qs.PFLOOR = True
    qs.EPS2 = 1e-20
    ev, steps, sweeps = qs.ql(dg, e2, 0, wave=1)
    ev2, steps_w, sweeps_w = qs.ql(dg, e2, 1, wave=64)
    return steps / dg.shape[0], sweeps / dg.shape[0], steps_w / (dg.shape[0] // 64), sweeps_w / (dg.shape[0] // 64)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    f = _lapack()
    x = qs.synth(4096, D, 1234)
    L = np.linalg.cholesky(x)
    Linv = np.linalg.inv(L)
    tot = np.zeros(3)
    npb = 0
    ql = np.zeros(4)
    worst = 0.0
    per_problem = []
    for i in range(rows):
        M = np.einsum("ab,nbc,dc->nad", Linv[i], x, Linv[i])
        M = 0.5 * (M + M.transpose(0, 2, 1))
        dg, e2 = qs.tridiag(M)
        q, e = chol_qe(dg, e2)
        ev, cnt = dlasq2_counts(q, e, f)
        ref = np.sort(np.linalg.eigvalsh(M), axis=1)
        worst = max(worst, np.max(np.abs(np.sort(ev, 1) - ref) / ref))
        tot += cnt.sum(0)
        per_problem.append(cnt[:, 1])
        npb += len(cnt)
        ql += np.array(ql_lone_lane_steps(dg, e2))
    ql /= rows
    sweeps, ndiv, fail = tot / npb
    pp = np.concatenate(per_problem)
    wave_max = pp.reshape(-1, 64).max(1).mean()
    print(f"problems: {npb} (d = {D}, benchmark distribution)")
    print(f"dlasq2 per problem: {sweeps:.1f} dqds sweeps, {ndiv:.1f} inner steps (divisions), {fail:.1f} % of shifts failed; max rel eig err {worst:.1e}")
    print(f"dlasq2, 64 problems in lock step at the granularity of a whole solve (max over the wave): {wave_max:.1f} steps")
    print(f"root-free QL of the kernel, a lane alone: {ql[0]:.1f} steps, {ql[1]:.1f} sweeps; wave of 64 with look-ahead: {ql[2]:.1f} steps, {ql[3]:.1f} sweeps")
    dq, qlc = 9.0, 19.0
    print(f"issue slots per problem, no lock-step tax: dqds {dq * ndiv:.0f} (+ ~12 per sweep of shift logic = {dq * ndiv + 12 * sweeps:.0f}, + ~70 for the"
          f" Cholesky of T) vs QL {qlc * ql[0]:.0f} (+ 26 per sweep = {qlc * ql[0] + 26 * ql[1]:.0f})")
    print(f"in lock step: QL today {qlc * ql[2] + 26 * ql[3]:.0f}; dqds with the SAME 1.23x tax {1.23 * (dq * ndiv + 12 * sweeps) + 70:.0f}, at whole-solve granularity"
          f" {dq * wave_max + 12 * sweeps * wave_max / ndiv + 70:.0f}")


if __name__ == "__main__":
    main()
