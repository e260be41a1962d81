"""Soak of the two-wave single-launch solve against the one-wave form (gabo_spd_tr_two_waves): random surrogates (d = 2 ... 6, 3 ... 55 training points, EI or posterior
mean, Gaussian or Laplace), random eigenvalue bounds (none / max / min / box; plain or strict), random starts and restart counts - candidates, values and iteration counts
must agree bit for bit.    python tools/soak_two_waves.py [--cases 200] [--seed 0]"""
import argparse
import ctypes
import functools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _lib, manifolds, models, ops                                                       # noqa: E402
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel, SpdAffineInvariantLaplaceKernel  # noqa: E402
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions                    # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold                    # noqa: E402
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut                               # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,         # noqa: E402
                                                            vector_to_symmetric_matrix_mandel_torch)
from oracle import spd as ospd                                                                               # noqa: E402

DEV = "cuda:0"


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    lib = _lib.load()
    rng = np.random.default_rng(a.seed)
    ops.set_error_checking(False)
    ran = same = iters_total = 0
    hits = misses = 0
    for case in range(a.cases):
        d = int(rng.integers(2, 7))
        n = int(rng.integers(3, min(56, int(lib.gabo_spd_acq_max_train(d)) + 1)))
        R = int(rng.choice([1, 7, 48, 130, 512]))
        q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
        Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.2, 3.0, (n, d)), q)
        X = ospd.symmetric_matrix_to_vector_mandel(0.5 * (Xm + Xm.transpose(0, 2, 1)))
        y = np.log(np.linalg.eigvalsh(Xm)).sum(1) ** 2 + 0.1 * rng.standard_normal(n)
        kern = (SpdAffineInvariantGaussianKernel if rng.random() < 0.8 else SpdAffineInvariantLaplaceKernel)(beta_min=float(rng.uniform(0.2, 1.0)))
        gp = models.ExactGP(t(X), t(y), kern, outputscale=float(rng.uniform(0.5, 2.0)), noise=10.0 ** rng.uniform(-3, -1))
        acq = (models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False) if rng.random() < 0.75 else models.PosteriorMean(gp, maximize=False))
        q = np.linalg.qr(rng.standard_normal((R, d, d)))[0]
        P = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.3, 2.8, (R, d)), q)
        x0 = ops.matrix_to_mandel(t(0.5 * (P + P.transpose(0, 2, 1))))[:, None]
        kind = rng.choice(["none", "max", "min", "box"])
        cons = []
        if kind in ("max", "box"):
            cons.append(functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=float(rng.uniform(1.5, 3.0))))
        if kind in ("min", "box"):
            cons.append(functools.partial(scut.min_eigenvalue_constraint_torch, minimum_eigenvalue=float(rng.uniform(0.2, 0.5))))
        strict = bool(cons) and rng.random() < 0.4
        maxiter = int(rng.choice([3, 12, 40]))
        out = []
        for flag in (1, 0):
            lib.gabo_spd_tr_two_waves(flag)
            solver = BatchedTrustRegions(mingradnorm=1e-5, maxiter=maxiter, strict_constraints=strict)
            c, v = gen_candidates_manifold(x0, acq, manifolds.PositiveDefinite(d), solver, vector_to_symmetric_matrix_mandel_torch,
                                           symmetric_matrix_to_vector_mandel_torch, inequality_constraints=cons or None, approx_hessian=True, options={})
            out.append((c.cpu().numpy(), v.cpu().numpy(), solver.log["per_restart_iterations"].cpu().numpy(), "one_launch_solve" in solver.log))
        lib.gabo_spd_tr_two_waves(1)
        h, m = ctypes.c_longlong(0), ctypes.c_longlong(0)
        lib.gabo_spd_tr_two_waves_counters(ctypes.byref(h), ctypes.byref(m), 1)
        hits, misses = hits + h.value, misses + m.value
        ok = all(np.array_equal(x, z, equal_nan=True) for x, z in zip(out[0][:3], out[1][:3]))
        ran += 1
        same += ok
        iters_total += int(out[0][2].sum())
        tag = f"S^{d}_++ n={n} R={R} {type(kern).__name__[17:-6]} {type(acq).__name__} cons={kind} strict={strict} maxiter={maxiter}"
        print(f"{case:4d} {tag:90s} two waves ran: {h.value + m.value > 0}  {'same' if ok else 'DIFFERENT'}", flush=True)
    print(f"{same} of {ran} cases bit-identical; {iters_total} restart-iterations; speculated step in {hits} iterations, not in {misses}")


if __name__ == "__main__":
    main()
