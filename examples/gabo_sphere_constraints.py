#!/usr/bin/env python3
"""GaBO on the sphere under constraints, on the MI355X - the flows of the reference's
examples/bo_sphere/constrained_benchmark_examples/{gabo_sphere_equality_constraints, gabo_sphere_inequality_constraints,
gabo_sphere_bound_constraints}.py: the constraint callables, the constrained `manifold.rand` each of them installs, the solver each of them
selects - AugmentedLagrangeMethod(maxiter=200, inner_solver=TrustRegions(maxiter=200), gammas_fact=0.05) by default for the first two
(their `solver_name = 'ALM'`), ConstrainedTrustRegions(maxiter=200) for the bounds (and for `--solver CTR`) - EI, 5 restarts / 100 raw samples.

    python examples/gabo_sphere_constraints.py --kind equality|inequality|bounds [--solver ALM|CTR] [--iters 10]

The augmented-Lagrangian method runs on all restarts in lock step (its inner trust-region solves batched, the acquisition through the fused
HIP evaluation; `options={"batched_alm": False}`: restart by restart on the host, as the reference drives it, manifold_optimize.py:207-220) -
both reproduce the reference's own outer iterates (tests/test_gpu_alm.py).  ConstrainedTrustRegions runs all restarts in lock step on the device.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import manifolds, models                                                           # noqa: E402
from gabotorch_amd._compat import ScaleKernel                                                          # noqa: E402
from gabotorch_amd.BO_test_functions.test_functions import ackley_function_sphere                      # noqa: E402
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel                             # noqa: E402
from gabotorch_amd.manifold_optimization.augmented_Lagrange_method import AugmentedLagrangeMethod      # noqa: E402
from gabotorch_amd.manifold_optimization.constrained_trust_regions import ConstrainedTrustRegions      # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold              # noqa: E402
from gabotorch_amd.manifold_optimization.robust_trust_regions import TrustRegions                      # noqa: E402

BETA_MIN = {3: 6.5, 4: 2.0, 5: 1.2, 10: 0.6, 20: 0.35, 50: 0.21, 100: 0.21}     # gabo_sphere_equality_constraints.py:158-171


def constraints(kind, dim):
    """-> (equality constraints, inequality constraints, sampler of feasible points, feasibility test on a numpy point)"""
    if kind == "equality":                      # gabo_sphere_equality_constraints.py:100-118: the great circle x[1] = yc
        yc = 0.0

        def sample():
            x = np.random.randn(dim)
            idx = np.arange(dim) != 1
            x[1] = yc
            x[idx] = x[idx] / np.linalg.norm(x[idx]) * np.sqrt(1 - yc ** 2)
            return x
        return [lambda x: x[..., 1] - yc], None, sample, lambda p: abs(p[1] - yc) < 2e-3
    if kind == "inequality":                    # gabo_sphere_inequality_constraints.py:100-141: the cap of half-angle pi / 4 around e_0
        angle = np.pi / 4.0

        def domain(x):
            return angle - torch.acos(torch.clamp(x[..., 0], -1.0, 1.0))

        def sample():
            # (the reference draws around the LAST axis, :124-141, while its constraint is centred on the first one: its initial data
            # lie outside the domain it then enforces; here the draw is around the constraint's own centre)
            s, c = np.sin(angle), np.cos(angle)
            x = np.random.rand(dim)
            x[1:] = 2 * s * x[1:] - s
            x[0] = (1 - c) * x[0] + c
            if np.linalg.norm(x[1:]) > s:
                x[1:] = x[1:] / np.linalg.norm(x[1:]) * s
            x[0] = np.sqrt(max(1 - np.sum(x[1:] ** 2), 0.0))
            return x
        return None, [domain], sample, lambda p: np.arccos(np.clip(p[0], -1, 1)) < angle + 2e-3
    if kind == "bounds":                        # gabo_sphere_bound_constraints.py:94-131 (S^2 only)
        assert dim == 3
        xl, yl, yu, zl, zu = 0.0, -0.6, 0.6, -0.6, 0.6
        cons = [lambda x: x[..., 0] - xl, lambda x: x[..., 1] - yl, lambda x: yu - x[..., 1], lambda x: x[..., 2] - zl, lambda x: zu - x[..., 2]]

        def sample():
            while True:
                s = np.array([np.random.uniform(xl, 1.0), np.random.uniform(yl, yu), np.random.uniform(zl, zu)])
                s = s / np.linalg.norm(s)
                if s[0] > xl and yl < s[1] < yu and zl < s[2] < zu:
                    return s
        return None, cons, sample, lambda p: p[0] > xl - 2e-3 and yl - 2e-3 < p[1] < yu + 2e-3 and zl - 2e-3 < p[2] < zu + 2e-3
    raise ValueError(kind)


def run(kind="equality", solver_name=None, dim=3, iters=10, restarts=5, raw=100, seed=1234, device="cuda:0", verbose=True, alm_maxiter=200):
    np.random.seed(seed)
    torch.manual_seed(seed)
    man = manifolds.Sphere(dim)
    eqs, ineqs, sample, feasible = constraints(kind, dim)
    man.rand = sample                           # "Replace sample function of the manifold by the constrained sampling" (:118)
    objective = lambda x: ackley_function_sphere(x, man)      # noqa: E731
    x_data = torch.tensor(np.stack([man.rand() for _ in range(5)]), device=device)
    y_data = torch.cat([objective(x) for x in x_data]).reshape(-1).to(device)
    solver_name = solver_name or ("CTR" if kind == "bounds" else "ALM")
    if solver_name == "CTR":
        solver = ConstrainedTrustRegions(maxiter=200)
    else:
        solver = AugmentedLagrangeMethod(maxiter=alm_maxiter, inner_solver=TrustRegions(maxiter=200), gammas_fact=0.05)
    best = [float(y_data.min())]
    for it in range(iters):
        kern = ScaleKernel(SphereGaussianKernel(beta_min=BETA_MIN.get(dim, 0.6)), outputscale_prior=models.GammaPrior(2.0, 0.15))
        gp = models.SingleTaskGP(x_data, y_data, kern, noise_prior=models.GammaPrior(1.1, 0.05))
        models.fit_gpytorch_model(gp)
        acq = models.ExpectedImprovement(gp, best_f=float(y_data.min()), maximize=False)
        new_x = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None,
                                        equality_constraints=eqs, inequality_constraints=ineqs, options={"device": device})
        new_y = objective(new_x[0]).reshape(-1).to(device)
        x_data = torch.cat([x_data, new_x.detach()])
        y_data = torch.cat([y_data, new_y])
        best.append(float(y_data.min()))
        if verbose:
            print(f"Iteration {it}\t Best f {best[-1]:.6f}\t feasible {bool(feasible(new_x[0].cpu().numpy()))}")
    return x_data, y_data, best, feasible


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="equality", choices=["equality", "inequality", "bounds"])
    ap.add_argument("--solver", default=None, choices=["ALM", "CTR"])
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    run(a.kind, a.solver, a.dim, a.iters)
