#!/usr/bin/env python3
"""Where one GaBO iteration spends its time (examples/gabo_spd.py flow, S^5_++, 512 restarts): surrogate fit, acquisition sweep, objective.
    python tools/bo_iteration_breakdown.py [--dim 5] [--iters 30]"""
import argparse
import functools
import os
import sys
import time
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--restarts", type=int, default=512)
    ap.add_argument("--raw", type=int, default=1024)
    a = ap.parse_args()
    dev = "cuda:0"
    np.random.seed(0)
    torch.manual_seed(0)
    man = manifolds.PositiveDefinite(a.dim)
    man.min_eig, man.max_eig = 0.001, 5.0
    man.rand = types.MethodType(spd_sample, man)
    objective = lambda x: ackley_function_spd(x, man)                  # noqa: E731
    con = functools.partial(max_eigenvalue_constraint_torch, maximum_eigenvalue=man.max_eig)
    x = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(man.rand()) for _ in range(5)]), device=dev)
    y = torch.cat([objective(v) for v in x]).reshape(-1).to(dev)
    solver = BatchedTrustRegions(mingradnorm=1e-4, maxiter=100)
    ops.set_error_checking(False)
    t = {"fit": [], "sweep": [], "objective": []}
    sync = torch.cuda.synchronize
    for it in range(a.iters):
        sync(); t0 = time.perf_counter()
        kern = ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=0.25), outputscale_prior=models.GammaPrior(2.0, 0.15))
        gp = models.SingleTaskGP(x, y, kern, noise_prior=models.GammaPrior(1.1, 0.05))
        models.fit_gpytorch_model(gp)
        sync(); t1 = time.perf_counter()
        acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
        nx = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=a.restarts, raw_samples=a.raw, bounds=None,
                                     options={"device": dev, "device_rand": True}, inequality_constraints=[con],
                                     pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                     post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
        sync(); t2 = time.perf_counter()
        ny = objective(nx[0]).reshape(-1).to(dev)
        x, y = torch.cat([x, nx.detach()]), torch.cat([y, ny])
        sync(); t3 = time.perf_counter()
        t["fit"].append(t1 - t0); t["sweep"].append(t2 - t1); t["objective"].append(t3 - t2)
    for k, v in t.items():
        v = np.array(v[3:]) * 1e3
        print(f"{k:10s} median {np.median(v):8.2f} ms   first {v[0]:8.2f}   last {v[-1]:8.2f}")
    print("best f", float(y.min()), "n_train", len(y))


if __name__ == "__main__":
    main()
