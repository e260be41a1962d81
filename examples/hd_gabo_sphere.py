#!/usr/bin/env python3
"""High-dimensional GaBO on the sphere through nested spheres (the flow of the reference's examples/hd_gabo_sphere.py:80-230) on the
MI355X: a GP with the NestedSphereGaussianKernel is fitted on the high-dimensional observations with fit_gpytorch_manifold (the
nested-sphere AXES are learnt on their spheres together with the Euclidean hyper-parameters), the data are projected to the latent
sphere, a latent GP with the SphereGaussianKernel and the same hyper-parameters is built, the distances to the axes of the
reconstruction are optimised (optimize_reconstruction_parameters_nested_sphere), EI is maximised on the LATENT sphere with stock
trust regions, and the winner is lifted back with projection_from_subsphere_to_sphere.  The objective is the Ackley function of the
reference evaluated on a fixed nested subsphere (nested_test_functions_sphere.py semantics: axes e_1, distances pi/4).

    python examples/hd_gabo_sphere.py [--dim 5] [--latent 3] [--iters 10]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import manifolds, models                                                                       # noqa: E402
from gabotorch_amd._compat import ScaleKernel                                                                      # noqa: E402
from gabotorch_amd.BO_test_functions.test_functions import ackley_function_sphere                                  # noqa: E402
from gabotorch_amd.kernel_utils.kernels_nested_sphere import NestedSphereGaussianKernel                            # noqa: E402
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel                                         # noqa: E402
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions                          # noqa: E402
from gabotorch_amd.manifold_optimization.conjugate_gradient import ConjugateGradient                               # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_gp_fit import fit_gpytorch_manifold                              # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold                          # noqa: E402
from gabotorch_amd.nested_mappings.nested_spheres_optimization import optimize_reconstruction_parameters_nested_sphere   # noqa: E402
from gabotorch_amd.nested_mappings.nested_spheres_utils import (projection_from_sphere_to_subsphere,               # noqa: E402
                                                                projection_from_subsphere_to_sphere)

BETA_MIN = {3: 6.5, 4: 2.0, 5: 1.2, 6: 1.0, 11: 0.6, 21: 0.35, 51: 0.21, 101: 0.21}        # hd_gabo_sphere.py:116-131


def run(dim=5, latent=3, iters=10, restarts=5, raw=100, seed=1234, device="cuda:0", verbose=True, fit_iters=50):
    np.random.seed(seed)
    torch.manual_seed(seed)
    big, small = manifolds.Sphere(dim), manifolds.Sphere(latent)
    # the test function lives on a fixed nested subsphere: axes e_1 of each level, distances pi/4 (:96-103)
    axes_test = [torch.zeros(dim - k, dtype=torch.float64, device=device) for k in range(dim - latent)]
    for a in axes_test:
        a[0] = 1.0
    dist_test = [torch.tensor([[np.pi / 4]], dtype=torch.float64) for _ in range(dim - latent)]

    def objective(x):
        z = projection_from_sphere_to_subsphere(x.reshape(1, -1), axes_test, dist_test)[-1]
        return ackley_function_sphere(z[0], small)

    x_data = torch.tensor(np.stack([big.rand() for _ in range(5)]), device=device)
    y_data = torch.cat([objective(x) for x in x_data]).reshape(-1).to(device)
    k_fct = ScaleKernel(NestedSphereGaussianKernel(dim, latent, beta_min=BETA_MIN.get(latent, 0.6)),
                        outputscale_prior=models.GammaPrior(2.0, 0.15)).double()
    solver = BatchedTrustRegions()                                                   # pyman_solvers.TrustRegions() (:166-167)
    best = [float(y_data.min())]
    for it in range(iters):
        model = models.SingleTaskGP(x_data, y_data, k_fct, noise_prior=models.GammaPrior(1.1, 0.05))
        fit_gpytorch_manifold(model, solver=ConjugateGradient(maxiter=fit_iters), nb_init_candidates=20)           # (:176)
        axes = [a.detach().clone().to(device) for a in k_fct.base_kernel.axes]
        dists = [d.clone() for d in k_fct.base_kernel.distances_to_axis]
        z_data = projection_from_sphere_to_subsphere(x_data, axes, dists)[-1]
        latent_kernel = SphereGaussianKernel(beta_min=BETA_MIN.get(latent, 0.6)).double()
        latent_kernel.beta = k_fct.base_kernel.beta.detach().clone()                 # same hyper-parameters (:186-188)
        gp = models.ExactGP(z_data, y_data, latent_kernel, outputscale=float(k_fct.outputscale.detach()), noise=float(model.noise.detach()),
                            mean=float(model.mean_constant.detach()))
        rec_dists = optimize_reconstruction_parameters_nested_sphere(x_data, z_data, axes, ConjugateGradient(maxiter=100),
                                                                     nb_init_candidates=30)                      # (:198-200)
        acq = models.ExpectedImprovement(gp, best_f=float(y_data.min()), maximize=False)
        z_new = joint_optimize_manifold(acq, small, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None,
                                        options={"device": device})
        x_new = projection_from_subsphere_to_sphere(z_new, axes, rec_dists)[-1]
        x_new = x_new / x_new.norm(dim=-1, keepdim=True)
        y_new = objective(x_new[0]).reshape(-1).to(device)
        x_data = torch.cat([x_data, x_new.detach()])
        y_data = torch.cat([y_data, y_new])
        best.append(float(y_data.min()))
        if verbose:
            print(f"Iteration {it}\t Best f {best[-1]:.6f}")
    return x_data, y_data, best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=5)
    ap.add_argument("--latent", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    run(a.dim, a.latent, a.iters)
