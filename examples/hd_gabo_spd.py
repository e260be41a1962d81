#!/usr/bin/env python3
"""High-dimensional GaBO on S^D_++ through a nested S^d_++ (the data flow of the reference's examples/hd_gabo_spd.py) on the MI355X:
observations live on S^D_++; they are projected to the latent S^d_++ with Y = W^T X W (gabo_spd_project), a GP with the
affine-invariant kernel is fitted on the latent points, EI is maximised ON THE LATENT MANIFOLD with the strict constrained trust
regions (eigenvalue box, FD Hessian), and the winner is lifted back with projection_from_nested_spd_to_spd.

Scope note: the reference also LEARNS the projection W (GP fit on a product manifold with pymanopt's conjugate gradient) and the
reconstruction parameters (augmented Lagrangian); those host-side optimisers are outside this repository's hot path,
so W, the bottom block and the contraction are fixed here.

    python examples/hd_gabo_spd.py [--dim 5] [--latent 2] [--iters 10]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import manifolds, models, ops                                                                 # noqa: E402
from gabotorch_amd._compat import ScaleKernel                                                                     # noqa: E402
from gabotorch_amd.BO_test_functions.test_functions import rosenbrock_function_spd                                # noqa: E402
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel                               # noqa: E402
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions                         # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold                         # noqa: E402
from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd                      # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_constraints_utils_torch import (max_eigenvalue_constraint_torch,          # noqa: E402
                                                                        min_eigenvalue_constraint_torch)
from gabotorch_amd.Riemannian_utils.spd_utils import spd_sample, symmetric_matrix_to_vector_mandel                # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,              # noqa: E402
                                                            vector_to_symmetric_matrix_mandel_torch)


def run(dim=5, latent=2, iters=10, restarts=5, raw=100, seed=1234, device="cuda:0", verbose=True):
    np.random.seed(seed)
    torch.manual_seed(seed)
    big = manifolds.PositiveDefinite(dim)
    big.min_eig, big.max_eig = 0.1, 5.0
    big.rand = types.MethodType(spd_sample, big)
    small = manifolds.PositiveDefinite(latent)
    small.min_eig, small.max_eig = 0.1, 5.0
    small.rand = types.MethodType(spd_sample, small)
    R = np.linalg.qr(np.random.randn(dim, dim))[0]
    W = torch.tensor(R[:, :latent], device=device)
    V = torch.tensor(R[:, latent:], device=device)
    bottom = torch.eye(dim - latent, dtype=torch.float64, device=device)
    contraction = torch.zeros(latent, dim - latent, dtype=torch.float64, device=device)
    objective = lambda x: rosenbrock_function_spd(x, big)          # noqa: E731  evaluated on the HIGH-dimensional manifold
    x_data = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(big.rand()) for _ in range(5)]), device=device)
    y_data = torch.cat([objective(x) for x in x_data]).reshape(-1).to(device)
    cons = [lambda x: max_eigenvalue_constraint_torch(x, small.max_eig), lambda x: min_eigenvalue_constraint_torch(x, small.min_eig)]
    solver = BatchedTrustRegions(mingradnorm=2e-4, maxiter=100, minstepsize=1e-4, strict_constraints=True)   # hd_gabo_spd.py:194
    ops.set_error_checking(False)
    best = [float(y_data.min())]
    for it in range(iters):
        z_data = ops.spd_project(x_data, W)                                          # latent Mandel vectors, one launch
        kern = ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=0.6), outputscale_prior=models.GammaPrior(2.0, 0.15))
        gp = models.SingleTaskGP(z_data, (y_data - y_data.mean()) / (y_data.std() + 1e-12), kern, noise_prior=models.GammaPrior(1.1, 0.05))
        models.fit_gpytorch_model(gp)
        acq = models.ExpectedImprovement(gp, best_f=float(gp.train_y.min()), maximize=False)
        z_new = joint_optimize_manifold(acq, small, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None,
                                        options={"device": device}, inequality_constraints=cons,
                                        pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                        post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
        x_new_mat = projection_from_nested_spd_to_spd(vector_to_symmetric_matrix_mandel_torch(z_new[0]), W, V, bottom, contraction)
        x_new = symmetric_matrix_to_vector_mandel_torch(x_new_mat)[None]
        y_new = objective(x_new[0]).reshape(-1).to(device)
        x_data = torch.cat([x_data, x_new.detach()])
        y_data = torch.cat([y_data, y_new])
        best.append(float(y_data.min()))
        if verbose:
            print(f"Iteration {it}\t Best f {best[-1]:.6f}")
    ops.set_error_checking(True)
    return x_data, y_data, best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=5)
    ap.add_argument("--latent", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    run(a.dim, a.latent, a.iters)
