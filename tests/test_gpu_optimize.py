"""Acquisition maximisation on the MI355X: HIP kernels + HIP manifold operations under the lock-step trust regions, against
(a) the reference solvers' optima (tests/golden/trust_regions.npz) and (b) the same pipeline on the torch-CPU stand-ins."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, manifolds, models, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedProblem, BatchedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,
                                                            vector_to_symmetric_matrix_mandel_torch)
from oracle import spd as ospd
from tests._cpu_manifolds import CpuSpd, CpuSphere, _sym

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


@pytest.mark.parametrize("n", [3, 5])
@pytest.mark.parametrize("approx", [False, True])
def test_sphere_hip_path_matches_reference_optima(golden, n, approx):
    g = golden("trust_regions.npz")
    Y, w, beta = t(g[f"sph{n}_Y"]), t(g[f"sph{n}_w"]), float(g[f"sph{n}_beta"])
    cost = lambda x: -(ops.sphere_kernel(x, Y, beta) * w).sum(-1)      # noqa: E731  R x 12 kernel strip per call
    x = BatchedTrustRegions().solve(BatchedProblem(manifolds.Sphere(n), cost, approx_hessian=approx), t(g[f"sph{n}_x0"]))
    key = "fd" if approx else "exact"
    np.testing.assert_allclose(cost(x).cpu().numpy(), g[f"sph{n}_{key}_f"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(x.cpu().numpy(), g[f"sph{n}_{key}_x"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("d", [2, 3])
def test_spd_hip_path_matches_reference_optima(golden, d):
    g = golden("trust_regions.npz")
    Ym = t(ospd.symmetric_matrix_to_vector_mandel(g[f"spd{d}_Y"]))
    w, beta = t(g[f"spd{d}_w"]), float(g[f"spd{d}_beta"])

    def cost(x):                                     # x: R x d x d matrices (the solver's representation)
        v = symmetric_matrix_to_vector_mandel_torch(x)
        return -(ops.spd_ai_kernel(v, Ym, beta) * w).sum(-1)
    man = manifolds.PositiveDefinite(d)
    ops.set_error_checking(False)
    try:
        x = BatchedTrustRegions(mingradnorm=1e-4, maxiter=100).solve(BatchedProblem(man, cost, approx_hessian=True), t(g[f"spd{d}_x0"]))
        np.testing.assert_allclose(cost(x).cpu().numpy(), g[f"spd{d}_fd_f"], rtol=1e-6)
        np.testing.assert_allclose(x.cpu().numpy(), g[f"spd{d}_fd_x"], rtol=0, atol=2e-3)
        mx = float(g[f"spd{d}_maxeig"])
        xc = BatchedTrustRegions(mingradnorm=1e-4, maxiter=100).solve(
            BatchedProblem(man, cost, approx_hessian=True), t(g[f"spd{d}_con_x0"]),
            ineq_constraints=[lambda x: scut.max_eigenvalue_constraint_torch(x, mx)])
        np.testing.assert_allclose(cost(xc).cpu().numpy(), g[f"spd{d}_con_f"], rtol=2e-3)
    finally:
        ops.set_error_checking(True)


class _CpuSpdKernel:
    """torch-CPU statement of exp(-beta d_AI^2) on Mandel inputs (test stand-in, differentiable by autograd)."""

    def __init__(self, beta):
        self.beta = beta

    def forward(self, x1, x2, **kw):
        a = torch.tensor(ospd.vector_to_symmetric_matrix_mandel(x1.detach().numpy())) if not x1.requires_grad else _mandel_to_mat(x1)
        b = torch.tensor(ospd.vector_to_symmetric_matrix_mandel(x2.detach().numpy())) if not x2.requires_grad else _mandel_to_mat(x2)
        L = torch.linalg.cholesky(a)
        Li = torch.linalg.inv(L)
        m = Li.unsqueeze(-3) @ b.unsqueeze(-4) @ Li.transpose(-1, -2).unsqueeze(-3)
        lam = torch.linalg.eigvalsh(_sym(m))
        return torch.exp(-self.beta * ((torch.log(lam) ** 2).sum(-1) + 1e-15))


def _mandel_to_mat(v):
    d = ospd.mandel_dim(v.shape[-1])
    r, c = ospd.mandel_index(d)
    scale = torch.tensor(np.where(r == c, 1.0, 1 / 2 ** 0.5))
    m = torch.zeros(v.shape[:-1] + (d, d), dtype=v.dtype)
    m[..., r, c] = v * scale
    m[..., c, r] = v * scale
    return m


def _mat_to_mandel(m):
    d = m.shape[-1]
    r, c = ospd.mandel_index(d)
    scale = torch.tensor(np.where(r == c, 1.0, 2 ** 0.5))
    return 0.5 * (m[..., r, c] + m[..., c, r]) * scale


def test_joint_optimize_spd_gp_ei_matches_cpu_pipeline():
    """Config-4-shaped sweep (GP + EI on S^3_++, max-eigenvalue constraint, FD Hessian, spd_sample as manifold.rand) on the HIP
    path vs the identical pipeline on torch-CPU stand-ins, same seeds => same initial conditions => same optimum."""
    d, n_train, R = 3, 12, 8
    rng = np.random.default_rng(0)
    q = np.linalg.qr(rng.standard_normal((n_train, d, d)))[0]
    Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.2, 3.0, (n_train, d)), q)
    X = ospd.symmetric_matrix_to_vector_mandel(0.5 * (Xm + Xm.transpose(0, 2, 1)))
    y = np.log(np.linalg.eigvalsh(Xm)).sum(1) ** 2 + 0.1 * rng.standard_normal(n_train)
    beta = 0.5 + np.log(2)
    results = {}
    for where in ("hip", "cpu"):
        np.random.seed(7)
        torch.manual_seed(7)
        if where == "hip":
            man = manifolds.PositiveDefinite(d)
            kern = SpdAffineInvariantGaussianKernel(beta_min=0.5)
            assert abs(kern.beta.item() - beta) < 1e-6
            kern.beta = beta
            pre, post = vector_to_symmetric_matrix_mandel_torch, symmetric_matrix_to_vector_mandel_torch
            con = lambda x: scut.max_eigenvalue_constraint_torch(x, 3.5)        # noqa: E731
            tx, ty, opts = t(X), t(y), {"device": DEV}
        else:
            man = CpuSpd(d)
            kern = _CpuSpdKernel(kern_beta)
            pre, post = _mandel_to_mat, _mat_to_mandel
            con = lambda x: 3.5 - torch.linalg.eigvalsh(x)[..., -1]             # noqa: E731
            tx, ty, opts = torch.tensor(X), torch.tensor(y), {}
        kern_beta = float(kern.beta.item()) if where == "hip" else kern_beta
        man.min_eig, man.max_eig = 0.2, 3.0

        def rand(self=man):
            return ospd.spd_sample(d, self.min_eig, self.max_eig)
        man.rand = rand
        gp = models.ExactGP(tx, ty, kern, outputscale=1.0, noise=1e-2)
        acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
        ops.set_error_checking(False)
        best = joint_optimize_manifold(acq, man, BatchedTrustRegions(mingradnorm=1e-4, maxiter=100), q=1, num_restarts=R,
                                       raw_samples=64, bounds=None, options=opts, inequality_constraints=[con],
                                       pre_processing_manifold=pre, post_processing_manifold=post, approx_hessian=True)
        ops.set_error_checking(True)
        results[where] = (best.detach().cpu().numpy(), float(acq(best[None].to(tx.device)).item()))
    (bh, vh), (bc, vc) = results["hip"], results["cpu"]
    assert bh.shape == (1, d * (d + 1) // 2)
    np.testing.assert_allclose(vh, vc, rtol=1e-5)                         # acquisition optimum: BASELINE.json tolerance
    np.testing.assert_allclose(bh, bc, rtol=0, atol=2e-3)
    assert np.linalg.eigvalsh(ospd.vector_to_symmetric_matrix_mandel(bh[0])).min() > 0


def test_joint_optimize_sphere_gp_ei_exact_hessian():
    """Config-1-shaped sweep on S^2 (stock trust regions, exact Hessian-vector products through the double-differentiable
    sphere kernel): the returned candidate is on the sphere and at least as good as every initial condition."""
    rng = np.random.default_rng(1)
    X = rng.standard_normal((20, 3)); X /= np.linalg.norm(X, axis=1, keepdims=True)
    y = np.arccos(np.clip(X[:, 0], -1, 1)) ** 2 + 0.05 * rng.standard_normal(20)
    kern = SphereGaussianKernel(beta_min=6.5)
    gp = models.ExactGP(t(X), t(y), kern, outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    np.random.seed(3)
    torch.manual_seed(3)
    solver = BatchedTrustRegions()
    best = joint_optimize_manifold(acq, manifolds.Sphere(3), solver, q=1, num_restarts=16, raw_samples=200, bounds=None,
                                   options={"device": DEV})
    assert best.shape == (1, 3) and abs(best.norm().item() - 1) < 1e-12
    raw = torch.tensor(np.stack([manifolds.Sphere(3).rand() for _ in range(500)]), device=DEV)[:, None]
    assert acq(best[None]).item() >= acq(raw).max().item() - 1e-12


def test_hip_graph_evaluations_match_eager():
    """options={"hip_graphs": True}: the acquisition value / gradient evaluations are captured once and replayed; same optimum."""
    from tools.sweep_bench import run_sweep
    _, best_e, val_e, log_e = run_sweep(DEV, num_restarts=32, raw_samples=256)
    _, best_g, val_g, log_g = run_sweep(DEV, num_restarts=32, raw_samples=256, hip_graphs=True)
    np.testing.assert_allclose(best_g.cpu().numpy(), best_e.cpu().numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(val_g, val_e, rtol=1e-10)
    assert int(log_g["iterations"]) == int(log_e["iterations"])          # same trust-region trajectory
