#!/usr/bin/env python3
"""GaBO on the sphere S^{n-1} (Ackley benchmark) on the MI355X - the flow of the reference's examples/gabo_sphere.py:69-286:
SphereGaussianKernel with the beta_min ladder, stock trust regions with EXACT Hessian-vector products (double backward
through the sphere kernel), EI, 5 restarts / 100 raw samples.

    python examples/gabo_sphere.py [--dim 3] [--iters 25]
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
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions              # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold              # noqa: E402

BETA_MIN = {3: 6.5, 4: 2.0, 5: 1.2, 11: 0.6, 21: 0.35, 51: 0.21, 101: 0.21}     # examples/gabo_sphere.py:115-128


def run(dim=3, iters=25, restarts=5, raw=100, seed=1234, device="cuda:0", verbose=True):
    np.random.seed(seed)
    torch.manual_seed(seed)
    man = manifolds.Sphere(dim)
    objective = lambda x: ackley_function_sphere(x, man)      # noqa: E731
    x_data = torch.tensor(np.stack([man.rand() for _ in range(5)]), device=device)
    y_data = torch.cat([objective(x) for x in x_data]).reshape(-1).to(device)
    solver = BatchedTrustRegions()                              # pyman_solvers.TrustRegions() (:151)
    best = [float(y_data.min())]
    for it in range(iters):
        kern = ScaleKernel(SphereGaussianKernel(beta_min=BETA_MIN.get(dim, 0.6)), outputscale_prior=models.GammaPrior(2.0, 0.15))
        gp = models.SingleTaskGP(x_data, y_data, kern, noise_prior=models.GammaPrior(1.1, 0.05))
        models.fit_gpytorch_model(gp)
        acq = models.ExpectedImprovement(gp, best_f=float(y_data.min()), maximize=False)
        new_x = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None,
                                        options={"device": device})
        new_y = objective(new_x[0]).reshape(-1).to(device)
        x_data = torch.cat([x_data, new_x.detach()])
        y_data = torch.cat([y_data, new_y])
        best.append(float(y_data.min()))
        if verbose:
            print(f"Iteration {it}\t Best f {best[-1]:.6f}")
    return x_data, y_data, best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--iters", type=int, default=25)
    a = ap.parse_args()
    run(a.dim, a.iters)
