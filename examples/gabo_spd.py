#!/usr/bin/env python3
"""GaBO on the SPD manifold S^d_++ (Ackley benchmark) on the MI355X - the flow of the reference's examples/gabo_spd.py:79-313
(same kernel, priors, constraint, solver settings and acquisition), with the kernel evaluations, their gradients and the
manifold operations running in libgabo_hip.so and the acquisition restarts advancing in lock step.

    python examples/gabo_spd.py [--dim 3] [--iters 15] [--restarts 5] [--raw 100]
"""
import argparse
import functools
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import manifolds, models, ops                                                            # noqa: E402
from gabotorch_amd._compat import ScaleKernel                                                                # noqa: E402
from gabotorch_amd.BO_test_functions.test_functions import ackley_function_spd                               # noqa: E402
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel                          # noqa: E402
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions                    # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold                    # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch       # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils import spd_sample, symmetric_matrix_to_vector_mandel           # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,         # noqa: E402
                                                            vector_to_symmetric_matrix_mandel_torch)

BETA_MIN = {2: 0.6, 3: 0.5, 5: 0.25, 7: 0.22, 10: 0.2, 12: 0.16}      # examples/gabo_spd.py:151-162


def run(dim=3, iters=15, restarts=5, raw=100, seed=1234, device="cuda:0", verbose=True, hip_graphs=False):
    np.random.seed(seed)
    torch.manual_seed(seed)
    man = manifolds.PositiveDefinite(dim)
    man.min_eig, man.max_eig = 0.001, 5.0
    man.rand = types.MethodType(spd_sample, man)                        # examples/gabo_spd.py:102
    objective = lambda x: ackley_function_spd(x, man)                  # noqa: E731
    constraint = functools.partial(max_eigenvalue_constraint_torch, maximum_eigenvalue=man.max_eig)     # (:136-138)
    x_data = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(man.rand()) for _ in range(5)]), device=device)
    y_data = torch.cat([objective(x) for x in x_data]).reshape(-1).to(device)
    beta_min = BETA_MIN.get(dim, 0.2)
    solver = BatchedTrustRegions(mingradnorm=1e-4, maxiter=100)          # ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100) (:183)
    ops.set_error_checking(False)
    best = [float(y_data.min())]
    for it in range(iters):
        kern = ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=beta_min), outputscale_prior=models.GammaPrior(2.0, 0.15))
        gp = models.SingleTaskGP(x_data, y_data, kern, noise_prior=models.GammaPrior(1.1, 0.05))
        models.fit_gpytorch_model(gp)
        acq = models.ExpectedImprovement(gp, best_f=float(y_data.min()), maximize=False)
        new_x = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None,
                                        options={"device": device, "hip_graphs": hip_graphs}, inequality_constraints=[constraint],
                                        pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                        post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
        new_y = objective(new_x[0]).reshape(-1).to(device)
        x_data = torch.cat([x_data, new_x.detach()])
        y_data = torch.cat([y_data, new_y])
        best.append(float(y_data.min()))
        if verbose:
            print(f"Iteration {it}\t Best f {best[-1]:.6f}")
    ops.set_error_checking(True)
    return x_data, y_data, best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--restarts", type=int, default=5)
    ap.add_argument("--raw", type=int, default=100)
    a = ap.parse_args()
    run(a.dim, a.iters, a.restarts, a.raw)
