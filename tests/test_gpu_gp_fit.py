"""Surrogate fit on product manifolds (f4): the marginal likelihood and its gradients with respect to every hyper-parameter -
including the nested-SPD projection matrix (Grassmannian) and the nested-sphere axes (spheres) - go through the HIP kernels."""
import numpy as np
import pytest
import torch

from gabotorch_amd import models
from gabotorch_amd._compat import ScaleKernel
from gabotorch_amd.kernel_utils.kernels_nested_sphere import NestedSphereGaussianKernel
from gabotorch_amd.kernel_utils.kernels_spd import NestedSpdLogEuclideanGaussianKernel
from gabotorch_amd.manifold_optimization.conjugate_gradient import ConjugateGradient
from gabotorch_amd.manifold_optimization.manifold_gp_fit import fit_gpytorch_manifold
from oracle import spd as ospd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand_spd(rng, n, d):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.3, 3.0, (n, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def test_mll_gradients_of_structured_parameters_by_finite_differences():
    rng = np.random.default_rng(0)
    D, dl, n = 5, 2, 14
    X = ospd.symmetric_matrix_to_vector_mandel(_rand_spd(rng, n, D))
    y = rng.standard_normal(n)
    kern = ScaleKernel(NestedSpdLogEuclideanGaussianKernel(D, dl)).double()
    gp = models.SingleTaskGP(torch.tensor(X, device=DEV), torch.tensor(y, device=DEV), kern)
    gp.marginal_log_likelihood().backward()
    W = kern.base_kernel.raw_projection_matrix
    g = W.grad.clone().numpy()
    num = np.zeros_like(g)
    h = 1e-6
    for idx in np.ndindex(g.shape):
        with torch.no_grad():
            W[idx] += h
            fp = float(gp.marginal_log_likelihood())
            W[idx] -= 2 * h
            fm = float(gp.marginal_log_likelihood())
            W[idx] += h
        num[idx] = (fp - fm) / (2 * h)
    np.testing.assert_allclose(g, num, rtol=1e-5, atol=1e-7)


def test_fit_learns_the_projection_matrix_on_the_grassmannian():
    rng = np.random.default_rng(3)
    D, dl, n = 5, 2, 30
    mats = _rand_spd(rng, n, D)
    W_true = np.linalg.qr(rng.standard_normal((D, dl)))[0]
    lat = np.einsum("da,ndc,cb->nab", W_true, mats, W_true)
    y = np.log(np.linalg.eigvalsh(lat)).sum(1)                         # depends on the data only through W_true^T X W_true
    y = (y - y.mean()) / y.std()
    X = ospd.symmetric_matrix_to_vector_mandel(mats)
    torch.manual_seed(0)
    np.random.seed(0)
    kern = ScaleKernel(NestedSpdLogEuclideanGaussianKernel(D, dl), outputscale_prior=models.GammaPrior(2.0, 0.15)).double()
    gp = models.SingleTaskGP(torch.tensor(X, device=DEV), torch.tensor(y, device=DEV), kern, noise_prior=models.GammaPrior(1.1, 0.05))
    before = float(-gp.marginal_log_likelihood().detach())
    _, info = fit_gpytorch_manifold(gp, solver=ConjugateGradient(maxiter=60), nb_init_candidates=40)
    W = kern.base_kernel.raw_projection_matrix.detach().numpy()
    np.testing.assert_allclose(W.T @ W, np.eye(dl), atol=1e-10)        # stayed on the manifold
    assert info["fopt"] < before - 0.05 and info["fopt"] <= info["init_cost"] + 1e-12
    # the learnt subspace is closer to the true one than a random subspace typically is (principal angles)
    overlap = np.linalg.svd(W_true.T @ W, compute_uv=False).min()
    assert overlap > 0.8, overlap


def test_fit_moves_nested_sphere_axes_on_their_spheres():
    rng = np.random.default_rng(4)
    dim, latent, n = 5, 3, 25
    X = rng.standard_normal((n, dim))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    y = X[:, 0] ** 2 - X[:, 1]
    y = (y - y.mean()) / y.std()
    torch.manual_seed(1)
    np.random.seed(1)
    kern = ScaleKernel(NestedSphereGaussianKernel(dim, latent, beta_min=0.5)).double()
    gp = models.SingleTaskGP(torch.tensor(X, device=DEV), torch.tensor(y, device=DEV), kern)
    before = float(-gp.marginal_log_likelihood().detach())
    a0 = [a.detach().clone() for a in kern.base_kernel.axes]
    _, info = fit_gpytorch_manifold(gp, solver=ConjugateGradient(maxiter=40), nb_init_candidates=20)
    assert info["fopt"] <= before + 1e-12
    for a, b in zip(kern.base_kernel.axes, a0):
        assert abs(float(a.norm()) - 1.0) < 1e-10
    assert any(float((a.detach() - b).abs().max()) > 1e-6 for a, b in zip(kern.base_kernel.axes, a0))
    # the fitted model predicts
    mean, var = gp.posterior(torch.tensor(X[:4], device=DEV))
    assert torch.isfinite(mean).all() and (var > -1e-9).all()
