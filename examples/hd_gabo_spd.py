#!/usr/bin/env python3
"""High-dimensional GaBO on S^D_++ through a nested S^d_++ (the data flow of the reference's examples/hd_gabo_spd.py:163-290) on the
MI355X: a GP with the nested log-Euclidean kernel is fitted on the HIGH-dimensional observations with fit_gpytorch_manifold -
which LEARNS the projection W on the Grassmannian together with the Euclidean hyper-parameters (conjugate gradients, 20 initial
candidates, as hd_gabo_spd.py:205) -, the data are projected with Y = W^T X W, a latent GP with the log-Euclidean kernel and the
same hyper-parameters is built on them, EI is maximised ON THE LATENT MANIFOLD with the strict constrained trust regions
(eigenvalue bounds stated in the original space, FD Hessian), and the winner is lifted back with projection_from_nested_spd_to_spd.

The reconstruction parameters (complement basis, bottom block, contraction) are optimised as in the reference
(optimize_reconstruction_parameters_nested_spd with the log-Euclidean cost, hd_gabo_spd.py:229-232; augmented Lagrangian + conjugate
gradients, --rec-iters 0 skips it and uses the orthogonal complement of W, C = I, K = 0).

    python examples/hd_gabo_spd.py [--dim 5] [--latent 2] [--iters 10]
"""
import argparse
import functools
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import manifolds, models, ops                                                                 # noqa: E402
from gabotorch_amd._compat import ScaleKernel                                                                     # noqa: E402
from gabotorch_amd.BO_test_functions.test_functions import rosenbrock_function_spd                                # noqa: E402
from gabotorch_amd.kernel_utils.kernels_spd import NestedSpdLogEuclideanGaussianKernel, SpdLogEuclideanGaussianKernel    # noqa: E402
from gabotorch_amd.manifold_optimization.conjugate_gradient import ConjugateGradient                              # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_gp_fit import fit_gpytorch_manifold                             # noqa: E402
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions                         # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold                         # noqa: E402
from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd                      # noqa: E402
from gabotorch_amd.nested_mappings.nested_spd_optimization import (min_log_euclidean_distance_reconstruction_cost,  # noqa: E402
                                                                   optimize_reconstruction_parameters_nested_spd)
from gabotorch_amd.nested_mappings.nested_spd_constraints_utils import (max_eigenvalue_nested_spd_constraint,    # noqa: E402
                                                                        min_eigenvalue_nested_spd_constraint,
                                                                        random_nested_spd_with_spd_eigenvalue_constraints)
from gabotorch_amd.Riemannian_utils.spd_utils import spd_sample, symmetric_matrix_to_vector_mandel                # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,              # noqa: E402
                                                            vector_to_symmetric_matrix_mandel_torch)


def run(dim=5, latent=2, iters=10, restarts=5, raw=100, seed=1234, device="cuda:0", verbose=True, fit_iters=50, rec_iters=6,
        timings=None):
    """timings: optional dict; the wall-clock seconds of the phases of every iteration are appended to its lists."""
    import time

    def lap(name, since):
        if timings is not None:
            torch.cuda.synchronize()
            timings.setdefault(name, []).append(time.perf_counter() - since)
        return time.perf_counter()
    np.random.seed(seed)
    torch.manual_seed(seed)
    big = manifolds.PositiveDefinite(dim)
    big.min_eig, big.max_eig = 0.1, 5.0
    big.rand = types.MethodType(spd_sample, big)
    small = manifolds.PositiveDefinite(latent)
    small.min_eig, small.max_eig = 0.1, 5.0
    small.rand = types.MethodType(spd_sample, small)
    bottom = torch.eye(dim - latent, dtype=torch.float64, device=device)
    contraction = torch.zeros(latent, dim - latent, dtype=torch.float64, device=device)
    objective = lambda x: rosenbrock_function_spd(x, big)          # noqa: E731  evaluated on the HIGH-dimensional manifold
    x_data = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(big.rand()) for _ in range(5)]), device=device)
    y_data = torch.cat([objective(x) for x in x_data]).reshape(-1).to(device)
    solver = BatchedTrustRegions(mingradnorm=2e-4, maxiter=100, minstepsize=1e-4, strict_constraints=True)   # hd_gabo_spd.py:194
    k_fct = ScaleKernel(NestedSpdLogEuclideanGaussianKernel(dim, latent), outputscale_prior=models.GammaPrior(2.0, 0.15)).double()
    ops.set_error_checking(False)
    best = [float(y_data.min())]
    for it in range(iters):
        tick = time.perf_counter()
        y_std = (y_data - y_data.mean()) / (y_data.std() + 1e-12)
        model = models.SingleTaskGP(x_data, y_std, k_fct, noise_prior=models.GammaPrior(1.1, 0.05))
        fit_gpytorch_manifold(model, solver=ConjugateGradient(maxiter=fit_iters), nb_init_candidates=20)           # :205
        W = k_fct.base_kernel.projection_matrix.detach().clone().to(device)
        z_data = ops.spd_project(x_data, W)                                          # latent Mandel vectors, one launch
        tick = lap("surrogate_fit", tick)
        if rec_iters > 0:                                                            # (:229-232)
            V, bottom, contraction = optimize_reconstruction_parameters_nested_spd(
                vector_to_symmetric_matrix_mandel_torch(x_data), vector_to_symmetric_matrix_mandel_torch(z_data), W,
                ConjugateGradient(maxiter=100), cost_function=min_log_euclidean_distance_reconstruction_cost, nb_init_candidates=20,
                maxiter=rec_iters)
        else:
            V = torch.linalg.svd(W, full_matrices=True)[0][:, latent:]              # orthonormal complement of span(W)
        tick = lap("reconstruction", tick)
        latent_kernel = SpdLogEuclideanGaussianKernel().double()
        latent_kernel.lengthscale = k_fct.base_kernel.lengthscale.detach().clone()   # same hyper-parameters (:217-219)
        # constraints and raw samples of the latent optimisation are stated in the ORIGINAL space (:239-256)
        small.rand = types.MethodType(functools.partial(random_nested_spd_with_spd_eigenvalue_constraints, random_spd_fct=big.rand,
                                                        projection_matrix=W), small)
        cons = [functools.partial(max_eigenvalue_nested_spd_constraint, maximum_eigenvalue=big.max_eig, projection_matrix=W,
                                  projection_complement_matrix=V, bottom_spd_matrix=bottom, contraction_matrix=contraction),
                functools.partial(min_eigenvalue_nested_spd_constraint, minimum_eigenvalue=big.min_eig, projection_matrix=W,
                                  projection_complement_matrix=V, bottom_spd_matrix=bottom, contraction_matrix=contraction)]
        gp = models.ExactGP(z_data, y_std, latent_kernel, outputscale=float(k_fct.outputscale.detach()), noise=float(model.noise.detach()),
                            mean=float(model.mean_constant.detach()))
        acq = models.ExpectedImprovement(gp, best_f=float(y_std.min()), maximize=False)
        z_new = joint_optimize_manifold(acq, small, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None,
                                        options={"device": device, "hip_graphs": True}, inequality_constraints=cons,
                                        pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                        post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
        tick = lap("latent_sweep", tick)
        x_new_mat = projection_from_nested_spd_to_spd(vector_to_symmetric_matrix_mandel_torch(z_new[0]), W, V, bottom, contraction)
        x_new = symmetric_matrix_to_vector_mandel_torch(x_new_mat)[None]
        y_new = objective(x_new[0]).reshape(-1).to(device)
        x_data = torch.cat([x_data, x_new.detach()])
        y_data = torch.cat([y_data, y_new])
        best.append(float(y_data.min()))
        lap("objective", tick)
        if verbose:
            print(f"Iteration {it}\t Best f {best[-1]:.6f}")
    ops.set_error_checking(True)
    return x_data, y_data, best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=5)
    ap.add_argument("--latent", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rec-iters", type=int, default=6)
    a = ap.parse_args()
    run(a.dim, a.latent, a.iters, rec_iters=a.rec_iters)
