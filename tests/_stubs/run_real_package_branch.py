"""Executed in a process of its own by tests/test_real_package_branch_*.py: installs the gpytorch / botorch shaped stand-ins, THEN imports
gabotorch_amd, and exercises every kernel class (and, with `gpu`, their forward and joint_optimize_manifold).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import real_package_stubs as stubs  # noqa: E402

gp, bo = stubs.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gabotorch_amd import _compat  # noqa: E402
from gabotorch_amd.plugin_api import botorch as api_botorch  # noqa: E402
from gabotorch_amd.plugin_api import gpytorch as api_gpytorch  # noqa: E402
from gabotorch_amd.kernel_utils import kernels_nested_sphere, kernels_spd, kernels_sphere  # noqa: E402

out = {"have_gpytorch": _compat.HAVE_GPYTORCH, "kernel_base_is_the_package_one": _compat.Kernel is gp.kernels.Kernel,
       "greater_than_is_the_package_one": _compat.GreaterThan is gp.constraints.GreaterThan,
       "plugin_api_reexports": api_gpytorch.kernels is gp.kernels and api_gpytorch.constraints is gp.constraints
       and api_botorch.acquisition is bo.acquisition and api_botorch.fit_gpytorch_model is bo.fit_gpytorch_model}

prior = gp.priors.GammaPrior(2.0, 0.15)
W = torch.linalg.qr(torch.randn(5, 2, dtype=torch.float64))[0]
made = {}
classes = [(kernels_spd.SpdAffineInvariantGaussianKernel, dict(beta_min=0.25, beta_prior=prior)),
           (kernels_spd.SpdAffineInvariantLaplaceKernel, dict(beta_min=0.25)),
           (kernels_spd.SpdFrobeniusGaussianKernel, {}), (kernels_spd.SpdLogEuclideanGaussianKernel, {}),
           (kernels_spd.NestedSpdAffineInvariantGaussianKernel, dict(dim=5, latent_dim=2, beta_min=0.25)),
           (kernels_spd.NestedSpdLogEuclideanGaussianKernel, dict(dim=5, latent_dim=2)),
           (kernels_sphere.SphereGaussianKernel, dict(beta_min=0.6, beta_prior=prior)), (kernels_sphere.SphereLaplaceKernel, dict(beta_min=0.6)),
           (kernels_nested_sphere.NestedSphereGaussianKernel, dict(dim=5, latent_dim=3, beta_min=0.6))]
for cls, kw in classes:
    try:
        k = cls(**kw)
    except TypeError:
        kw = {a: b for a, b in kw.items() if a not in ("dim", "latent_dim")}
        k = cls(**kw)
    assert isinstance(k, gp.kernels.Kernel), cls
    rec = {"params": sorted(n for n, _ in k.named_parameters())}
    if hasattr(k, "beta"):
        k.beta = 1.5
        rec["beta"] = float(k.beta.detach())
        assert isinstance(k.raw_beta_constraint, gp.constraints.GreaterThan)
    if getattr(k, "has_lengthscale", False):
        k.lengthscale = 0.8
        rec["lengthscale"] = float(k.lengthscale.detach())
    made[cls.__name__] = rec
out["kernels"] = made
scaled = gp.kernels.ScaleKernel(kernels_spd.SpdAffineInvariantGaussianKernel(beta_min=0.25), outputscale_prior=prior)
out["registration_calls"] = sorted({f"{c}.{m}({a})" for c, m, a in stubs.Module.calls})
out["scale_kernel_params"] = sorted(n for n, _ in scaled.named_parameters())

if len(sys.argv) > 1 and sys.argv[1] == "gpu":
    from gabotorch_amd import manifolds, models
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat
    from oracle import spd as ospd, sphere as osph
    dev = "cuda:0"
    rng = np.random.default_rng(3)
    q = np.linalg.qr(rng.standard_normal((12, 3, 3)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.3, 3.0, (12, 3)), q)
    xv = ospd.symmetric_matrix_to_vector_mandel(0.5 * (X + X.transpose(0, 2, 1)))
    k = kernels_spd.SpdAffineInvariantGaussianKernel(beta_min=0.25)
    k.beta = 0.9
    got = k(torch.tensor(xv, device=dev), torch.tensor(xv[:5], device=dev))        # gpytorch.kernels.Kernel.__call__ of the stand-in -> forward -> HIP
    out["spd_forward_err"] = float(np.abs(got.detach().cpu().numpy() - ospd.spd_ai_gaussian_kernel(xv, xv[:5], float(k.beta.detach().double()))).max())
    sx = rng.standard_normal((9, 4))
    sx /= np.linalg.norm(sx, axis=1, keepdims=True)
    ks = kernels_sphere.SphereGaussianKernel(beta_min=0.6)
    ks.beta = 1.1
    out["sphere_forward_err"] = float(np.abs(ks(torch.tensor(sx, device=dev)).detach().cpu().numpy() - osph.sphere_gaussian_kernel(sx, sx, float(ks.beta.detach().double()))).max())
    out["scaled_forward_ratio"] = float(((scaled(torch.tensor(xv, device=dev)) / scaled.base_kernel(torch.tensor(xv, device=dev))).mean() / scaled.outputscale).detach())
    # the maximiser driven by a FOREIGN acquisition object (the stand-in botorch's ExpectedImprovement over this package's GP posterior)
    y = rng.standard_normal(12)
    gpm = models.ExactGP(torch.tensor(xv, device=dev), torch.tensor(y, device=dev), k, outputscale=1.0, noise=1e-2)
    acq = bo.acquisition.ExpectedImprovement(gpm, best_f=float(y.min()), maximize=False)
    man = manifolds.PositiveDefinite(3)
    man.min_eig, man.max_eig = 0.3, 3.0
    np.random.seed(0)
    torch.manual_seed(0)
    best = joint_optimize_manifold(acq, man, BatchedTrustRegions(mingradnorm=1e-4, maxiter=20), q=1, num_restarts=4, raw_samples=16, bounds=None,
                                   options={"device": dev}, pre_processing_manifold=to_mat, post_processing_manifold=to_vec, approx_hessian=True)
    start = torch.tensor(xv[:1], device=dev)
    out["acq_at_optimum"], out["acq_at_a_training_point"] = float(acq(best[None].to(dev)).reshape(-1)[0]), float(acq(start[None]).reshape(-1)[0])
print(json.dumps(out))
