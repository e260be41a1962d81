#!/usr/bin/env python3
"""Golden best-of-restarts optimum of a config-4-sized EXPECTED-IMPROVEMENT sweep, produced by the reference's own solver (development
container only; needs /root/reference).  north_star: "acquisition optima within 1e-5" - this is the composed statement: the candidate
`joint_optimize_manifold` RETURNS (arg max over restarts, manifold_optimize.py:118-120) against the reference's best-of-restarts from the
same initial conditions.

  * surrogate: exact GP on 50 points of S^5_++ (eigenvalues U[1e-3, 5], seed 1234: the set of tools/sweep_bench.py / bench.py's config-4
    sweep), y = the reference's `ackley_function_spd` (BO_test_functions/test_functions_spd.py:14-69, manifold.log := the reference's own
    numpy log map), kernel exp(-beta d^2) with the reference's `affine_invariant_distance_torch` (spd_utils_torch.py:53-121), beta = 0.25 + ln 2,
    outputscale 1, noise 1e-2, constant mean = mean(y);
  * acquisition: expected improvement, maximize=False, botorch's formula [3P] stated here in torch: sigma = sqrt(clamp_min(var, 1e-9)),
    u = -(mu - best_f) / sigma, EI = sigma (phi(u) + u Phi(u));
  * solver: the reference's `ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100)` (examples/bo_spd/benchmark_examples/gabo_spd.py:183)
    with `problem._hess = get_hessianfd` (manifold_optimize.py:198-202) and the lambda_max <= 5 constraint of the example (:136-138),
    from 32 fixed starts drawn like the raw samples (eight of them on the constraint's edge), one after the other as manifold_optimize.py:207 does;
  * torch's DEFAULT dtype is float64 while the reference code runs (the "f64" convention of make_golden_tr_traces.py: the reference's eigenvalue
    buffer `torch.zeros(...)` at spd_utils_torch.py:108 is then double - the arithmetic the HIP path implements).

Stored: the training set, y, the hyper-parameters, the starts, and per restart the end point, its cost (-EI) and the iteration count.
"""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_tr as base  # noqa: E402  (installs the pymanopt Solver stand-in, imports the reference solvers)
from make_golden_tr_traces import traced  # noqa: E402  (subclass that records every outer iteration of the unmodified solver)

from BoManifolds.BO_test_functions import test_functions_spd as tf_spd  # noqa: E402
from BoManifolds.Riemannian_utils import spd_utils  # noqa: E402
from BoManifolds.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch  # noqa: E402

D, N_TRAIN, R = 5, 50, 32


def rand_spd(rng, k, d, lo, hi):
    q = np.linalg.qr(rng.standard_normal((k, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (k, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


class _LogMan:
    def __init__(self, n):
        self._n = n

    def log(self, b, x):      # pymanopt order: Log at `b` of `x`
        return np.real(spd_utils.logmap(x, b))


def main():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(1234)
    X = rand_spd(rng, N_TRAIN, D, 1e-3, 5.0)                       # the draw order of tools/sweep_bench.py: Q first, then the eigenvalues
    xm = spd_utils.symmetric_matrix_to_vector_mandel(X[0])
    Xv = np.stack([spd_utils.symmetric_matrix_to_vector_mandel(m) for m in X])
    man_log = _LogMan(D)
    y = np.array([np.asarray(tf_spd.ackley_function_spd(torch.tensor(v), man_log)).item() for v in Xv])
    beta = 0.25 + float(np.log(2.0))
    outputscale, noise, mean = 1.0, 1e-2, float(y.mean())
    best_f = float(y.min())

    Xt = torch.tensor(X)
    t0 = time.time()
    with torch.no_grad():
        Dtt = base.affine_invariant_distance_torch(Xt, Xt)
        K = outputscale * torch.exp(-beta * Dtt * Dtt) + noise * torch.eye(N_TRAIN)
        L = torch.linalg.cholesky(K)
        alpha = torch.cholesky_solve((torch.tensor(y) - mean).unsqueeze(-1), L).squeeze(-1)
    print(f"train-train distances: {time.time() - t0:.1f} s", flush=True)

    def neg_ei(x):                                                  # x: (5, 5) torch tensor (the reference's `cost`, manifold_optimize.py:177-185)
        xx = x[None].double()
        d = base.affine_invariant_distance_torch(xx, Xt)           # 1 x 50 (candidate first: the Cholesky factor is the candidate's)
        ks = outputscale * torch.exp(-beta * d * d)[0]
        dss = base.affine_invariant_distance_torch(xx, xx)
        kss = outputscale * torch.exp(-beta * dss * dss)[0, 0]
        mu = mean + ks @ alpha
        v = torch.linalg.solve_triangular(L, ks.unsqueeze(-1), upper=False).squeeze(-1)
        var = kss - (v * v).sum()
        sigma = var.clamp_min(1e-9).sqrt()
        u = -(mu - best_f) / sigma
        pdf = torch.exp(-0.5 * u * u) / np.sqrt(2.0 * np.pi)
        cdf = 0.5 * (1.0 + torch.erf(u / np.sqrt(2.0)))
        return -(sigma * (pdf + u * cdf))

    def precon(x, d):                                               # manifold_optimize.py:189-192
        if np.sum(d) == 0.0:
            d += 1e-30
        return d

    man = base.SpdMan(D)
    problem = base.Problem(manifold=man, cost=neg_ei, verbosity=0, arg=torch.Tensor(), precon=precon)
    problem._hess = types.MethodType(base.get_hessianfd, problem)
    con = [lambda x: max_eigenvalue_constraint_torch(x, 5.0)]
    # starts with the distribution of the raw samples (`spd_sample`, eigenvalues U[1e-3, 5]: gabo_spd.py:102, 200), the last eight with their
    # largest eigenvalue AT the bound's edge (4.99): restarts that begin on the constraint
    rng0 = np.random.default_rng(4321)
    x0 = rand_spd(rng0, R, D, 1e-3, 5.0)
    q = np.linalg.qr(rng0.standard_normal((8, D, D)))[0]
    lam = rng0.uniform(1e-3, 5.0, (8, D))
    lam[:, 0] = 4.99
    edge = np.einsum("nab,nb,ncb->nac", q, lam, q)
    x0[R - 8:] = 0.5 * (edge + edge.transpose(0, 2, 1))
    xs, fs, f0, nits = [], [], [], []
    for r in range(R):
        solver = traced(base.ConstrainedTrustRegions)(mingradnorm=1e-4, maxiter=100)
        solver.trace = []
        t0 = time.time()
        x = solver.solve(problem, x=x0[r].copy(), ineq_constraints=con)
        xs.append(np.asarray(x))
        fs.append(problem.cost(x))
        f0.append(problem.cost(x0[r]))
        nits.append(len(solver.trace))
        print(f"restart {r}: -EI {f0[-1]:.6e} -> {fs[-1]:.9e} in {nits[-1]} iterations, lambda_max {np.linalg.eigvalsh(xs[-1]).max():.6f} "
              f"({time.time() - t0:.1f} s)", flush=True)
    fs = np.array(fs)
    best = int(np.argmin(fs))
    print("best restart", best, "EI*", -fs[best])
    np.savez_compressed(os.path.join(HERE, "ei_optimum.npz"), X=X, Xv=Xv, y=y, beta=np.float64(beta), outputscale=np.float64(outputscale),
                        noise=np.float64(noise), mean=np.float64(mean), best_f=np.float64(best_f), x0=x0, x=np.stack(xs), f=fs,
                        f0=np.array(f0), nit=np.array(nits), best=np.int64(best), maxeig=np.float64(5.0), mandel_check=xm)
    torch.set_default_dtype(torch.float32)


if __name__ == "__main__":
    main()
