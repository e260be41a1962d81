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
        assert abs(float(a.detach().norm()) - 1.0) < 1e-10
    assert any(float((a.detach() - b).abs().max()) > 1e-6 for a, b in zip(kern.base_kernel.axes, a0))
    # the fitted model predicts
    mean, var = gp.posterior(torch.tensor(X[:4], device=DEV))
    assert torch.isfinite(mean).all() and (var > -1e-9).all()


# ----------------------------------------------------------------------------------- reconstruction of the nested-SPD mapping (f4)
def test_matrix_function_gradients_and_reconstruction_costs_golden(golden):
    """logm_torch / sqrtm_torch gradients, frobenius_distance_torch and the two reconstruction costs with their gradients w.r.t. the
    complement basis, the bottom block and the contraction, against vectors produced by the reference (autograd)."""
    from gabotorch_amd.nested_mappings import nested_spd_optimization as nso
    from gabotorch_amd.Riemannian_utils import spd_utils_torch as sut
    g = golden("reconstruction.npz")
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    sym = lambda a: 0.5 * (a + a.transpose(0, 2, 1))     # noqa: E731
    for name, fn in (("logm", sut.logm_torch), ("sqrtm", sut.sqrtm_torch)):
        a = T(g["mf_A"], True)
        y = fn(a)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{name}_val"], atol=1e-11)
        (y * T(g["mf_G"])).sum().backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), sym(g[f"{name}_grad"]), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(sut.frobenius_distance_torch(T(g["frob_x1"]), T(g["frob_x2"])).cpu().numpy(), g["frob_d"], rtol=1e-11)
    for tag in "ab":
        for name, fn in (("ai", nso.min_affine_invariant_distance_reconstruction_cost), ("le", nso.min_log_euclidean_distance_reconstruction_cost)):
            V, C, K = T(g[f"{tag}_V"], True), T(g[f"{tag}_C"], True), T(g[f"{tag}_K"], True)
            cost = fn(T(g[f"{tag}_X"]), T(g[f"{tag}_Y"]), T(g[f"{tag}_W"]), V, C, K)
            np.testing.assert_allclose(cost.item(), g[f"{tag}_{name}_cost"], rtol=2e-6)          # (the reference sums in float32)
            cost.backward()
            np.testing.assert_allclose(V.grad.cpu().numpy(), g[f"{tag}_{name}_gV"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(0.5 * (C.grad + C.grad.T).cpu().numpy(), 0.5 * (g[f"{tag}_{name}_gC"] + g[f"{tag}_{name}_gC"].T),
                                       rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(K.grad.cpu().numpy(), g[f"{tag}_{name}_gK"], rtol=1e-4, atol=1e-5)


def test_optimize_reconstruction_parameters_nested_spd():
    from gabotorch_amd.nested_mappings import nested_spd_optimization as nso
    from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd, projection_from_spd_to_nested_spd
    rng = np.random.default_rng(8)
    D, d, N = 4, 2, 8
    # data that ARE reconstructible: X = R [[Y, B], [B^T, C]] R^T with one (V, C, K)
    R = np.linalg.qr(rng.standard_normal((D, D)))[0]
    W, V0 = R[:, :d], R[:, d:]
    Y = _rand_spd(rng, N, d)
    C0 = _rand_spd(rng, 1, D - d)[0]
    K0 = rng.standard_normal((d, D - d))
    K0 = 0.5 * K0 / np.linalg.norm(K0)
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)   # noqa: E731
    X = projection_from_nested_spd_to_spd(T(Y), T(W), T(V0), T(C0), T(K0))
    np.testing.assert_allclose(projection_from_spd_to_nested_spd(X, T(W)).cpu().numpy(), Y, atol=1e-10)
    np.random.seed(5)
    torch.manual_seed(5)
    V, C, K = nso.optimize_reconstruction_parameters_nested_spd(X, T(Y), T(W), ConjugateGradient(maxiter=40), nb_init_candidates=30,
                                                                maxiter=12)
    log = nso.optimize_reconstruction_parameters_nested_spd.last_log
    assert log["final_cost"] < 0.25 * log["init_cost"], log
    assert float(torch.norm(T(W).T @ V)) < 5e-2                       # the constraint W^T V = 0 is approached by the multiplier method
    assert np.linalg.eigvalsh(C.cpu().numpy()).min() > 0 and float(torch.linalg.matrix_norm(K, 2)) < 1.0
    Xr = projection_from_nested_spd_to_spd(T(Y), T(W), V, C, K)
    assert np.linalg.eigvalsh(Xr.cpu().numpy()).min() > 0


@pytest.mark.parametrize("D,d,N,metric", [(4, 2, 8, "ai"), (9, 3, 6, "le"), (20, 2, 12, "le"), (20, 2, 12, "ai")])
def test_native_reconstruction_loop_follows_the_python_loop(D, d, N, metric):
    """gabo_nested_spd_reconstruction_solve (augmented Lagrangian + conjugate gradients in C++ around the fused launch) against the Python
    statement of the same algorithm (augmented_lagrange_method.py + conjugate_gradient.py + host_manifolds.py, themselves checked against
    analytic optima in tests/test_host_optimizers_cpu.py) from the same start: same iterates.  Short runs (rounding differences between the
    two eigen-solvers grow by a decade every ~5 conjugate-gradient iterations on this non-convex cost); the outer `min step size` test is
    switched off in both (it compares a Grassmann distance that is 1e-8-noise of the eigen-solver with 1e-10)."""
    from gabotorch_amd.nested_mappings import nested_spd_optimization as nso
    from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_spd_to_nested_spd
    rng = np.random.default_rng(100 + D)
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)   # noqa: E731
    X = T(_rand_spd(rng, N, D))
    W = T(np.linalg.qr(rng.standard_normal((D, d)))[0])
    Y = projection_from_spd_to_nested_spd(X, W)
    cost = nso.min_log_euclidean_distance_reconstruction_cost if metric == "le" else nso.min_affine_invariant_distance_reconstruction_cost
    out = {}
    for native in (False, True):
        np.random.seed(11)
        V, C, K = nso.optimize_reconstruction_parameters_nested_spd(X, Y, W, ConjugateGradient(maxiter=8), cost_function=cost, nb_init_candidates=10,
                                                                    maxiter=4, native=native, alm_options=dict(minstepsize=0.0))
        out[native] = (V.cpu().numpy(), C.cpu().numpy(), K.cpu().numpy(), dict(nso.optimize_reconstruction_parameters_nested_spd.last_log))
    py, nat = out[False], out[True]
    assert nat[3].get("native") and not py[3].get("native")
    assert nat[3]["iterations"] == py[3]["iterations"] == 4
    np.testing.assert_allclose(nat[3]["final_cost"], py[3]["final_cost"], rtol=1e-6)
    np.testing.assert_allclose(nat[3]["violation"], py[3]["violation"], rtol=1e-3, atol=1e-9)
    np.testing.assert_allclose(nat[3]["rho"], py[3]["rho"], rtol=1e-12)
    for a, b in zip(nat[:3], py[:3]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    assert nat[3]["launches"] <= nat[3]["evaluations"] <= 4 * nat[3]["launches"]          # (up to GABO_RECON_MAX_LOOKAHEAD step lengths per launch)


def test_nested_sphere_reconstruction_cost_and_optimiser(golden):
    from gabotorch_amd.nested_mappings import nested_spheres_optimization as nsso
    from gabotorch_amd.nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere
    g = golden("reconstruction.npz")
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    axes = [T(g["ns_axis0"]), T(g["ns_axis1"])]
    r = [T([[g["ns_r"][0]]], True), T([[g["ns_r"][1]]], True)]
    cost = nsso.min_error_reconstruction_cost(T(g["ns_x"]), T(g["ns_sub"]), axes, r)
    np.testing.assert_allclose(float(cost), g["ns_cost"], rtol=1e-10)
    cost.backward()
    np.testing.assert_allclose([float(r[0].grad), float(r[1].grad)], g["ns_grad"], rtol=1e-8)
    # optimiser: data lying exactly on nested small circles of radii (1.1, 0.9) are reconstructed with zero error at those radii
    rng = np.random.default_rng(12)
    true_r = [torch.tensor([[1.1]], dtype=torch.float64), torch.tensor([[0.9]], dtype=torch.float64)]
    z = rng.standard_normal((20, 3))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    from gabotorch_amd.nested_mappings.nested_spheres_utils import projection_from_subsphere_to_sphere
    x = projection_from_subsphere_to_sphere(T(z), axes, true_r)[-1]
    np.random.seed(2)
    found = nsso.optimize_reconstruction_parameters_nested_sphere(x, T(z), axes, ConjugateGradient(maxiter=200, mingradnorm=1e-9),
                                                                  nb_init_candidates=30)
    np.testing.assert_allclose([float(f) for f in found], [1.1, 0.9], atol=1e-5)
    assert float(nsso.min_error_reconstruction_cost(x, T(z), axes, found)) < 1e-9
    assert projection_from_sphere_to_subsphere(x, axes, found)[-1].shape == (20, 3)


@pytest.mark.parametrize("D,lat,N", [(5, 3, 9), (21, 3, 14), (51, 3, 11), (70, 2, 5)])
def test_fused_nested_sphere_reconstruction_launch(golden, D, lat, N):
    """gabo_nested_sphere_reconstruction (all levels of the lift, the distance to the data and the gradient w.r.t. the distances to the axes,
    P parameter sets per launch) against the level-by-level torch statement under autograd (min_error_reconstruction_cost, itself pinned on
    the reference's values in reconstruction.npz above), and on the golden fixture directly."""
    from gabotorch_amd import ops
    from gabotorch_amd.nested_mappings import nested_spheres_optimization as nsso
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    g = golden("reconstruction.npz")
    rec = ops.NestedSphereReconstruction(T(g["ns_x"]), T(g["ns_sub"]), [T(g["ns_axis0"]), T(g["ns_axis1"])])
    c, gr = rec.evaluate(np.asarray(g["ns_r"], dtype=np.float64))
    np.testing.assert_allclose(c, g["ns_cost"], rtol=1e-10)
    np.testing.assert_allclose(gr, g["ns_grad"], rtol=1e-8)
    rng = np.random.default_rng(D)
    L = D - lat
    axes = []
    for k in range(L):
        a = rng.standard_normal(D - k)
        axes.append(T(a / np.linalg.norm(a)))
    x = rng.standard_normal((N, D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    z = rng.standard_normal((N, lat))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    rec = ops.NestedSphereReconstruction(T(x), T(z), axes)
    P = 3
    r = rng.uniform(0.3, 2.8, size=(P, L))
    costs, grads = rec.evaluate(r)
    np.testing.assert_allclose(rec.evaluate(r, grad=False), costs, rtol=0, atol=0)
    for p in range(P):
        rt = [T([[v]], True) for v in r[p]]
        want = nsso.min_error_reconstruction_cost(T(x), T(z), axes, rt)
        want.backward()
        np.testing.assert_allclose(costs[p], float(want), rtol=1e-11)
        np.testing.assert_allclose(grads[p], [float(t_.grad) for t_ in rt], rtol=1e-8, atol=1e-10)
    c1, g1 = rec.evaluate(r[1])
    assert c1 == costs[1] and np.array_equal(g1, grads[1])


@pytest.mark.parametrize("D,lat,N", [(5, 3, 11), (21, 3, 16), (51, 3, 13), (101, 3, 4)])
def test_nested_sphere_chain_against_the_oracle(D, lat, N):
    """The all-levels launches against oracle/sphere.py (the numpy restatement of nested_spheres_utils.py with full rotation matrices, pinned on
    the reference's goldens by tests/test_oracle_golden.py): every level of the projection and of the lift, the reconstruction cost, and the
    nested kernel's Gram matrix through the surrogate objective's launch chain (its likelihood against oracle/gp.py)."""
    from gabotorch_amd import ops
    from gabotorch_amd.nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere, projection_from_subsphere_to_sphere
    from oracle import gp as ogp
    from oracle import sphere as osph
    rng = np.random.default_rng(7 * D)
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)   # noqa: E731
    L = D - lat
    axes_np = []
    for k in range(L):
        a = rng.standard_normal(D - k)
        axes_np.append(a / np.linalg.norm(a))
    r_np = rng.uniform(0.5, 2.5, L)
    x = rng.standard_normal((N, D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    axes, dists = [T(a) for a in axes_np], [torch.tensor([[v]], dtype=torch.float64) for v in r_np]
    down = projection_from_sphere_to_subsphere(T(x), axes, dists)
    want_down = osph.projection_from_sphere_to_subsphere(x, axes_np, r_np)
    for a, b in zip(down, want_down):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-9, atol=1e-11)
    z = want_down[-1]
    up = projection_from_subsphere_to_sphere(T(z), axes, dists)
    for a, b in zip(up, osph.projection_from_subsphere_to_sphere(z, axes_np, r_np)):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-9, atol=1e-11)
    rec = ops.NestedSphereReconstruction(T(x), T(z), axes)
    np.testing.assert_allclose(rec.evaluate(r_np, grad=False), osph.nested_sphere_reconstruction_cost(x, z, axes_np, r_np), rtol=1e-9)
    # the fit objective's chain: log likelihood of the nested kernel's Gram matrix at these axes
    import ctypes
    from gabotorch_amd import _lib
    lib = _lib.load()
    y = rng.standard_normal(N)
    beta, os_, noise, mean = 1.3, 0.8, 0.05, 0.1
    packed = ops.pack_nested_sphere_axes(axes_np, D)
    ws = torch.empty(lib.gabo_nested_sphere_fit_workspace_bytes(N, D, L), dtype=torch.uint8, device=DEV)
    pinned = torch.empty(2 * packed.size + L + 7, dtype=torch.float64).pin_memory()
    out = np.empty(7 + packed.size)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    xt, yt = T(x), T(y)
    _lib.check(lib.gabo_nested_sphere_fit_evaluate(xt.data_ptr(), yt.data_ptr(), ptr(packed), ptr(np.ascontiguousarray(r_np)), N, D, L, beta, os_, noise,
                                                   mean, 0, ptr(out), ws.data_ptr(), ws.numel(), pinned.data_ptr(), pinned.numel(), None), "fit_evaluate")
    gram = osph.nested_sphere_gaussian_kernel(x, x, axes_np, r_np, beta)
    ky = os_ * gram + noise * np.eye(N)
    resid = y - mean
    want_ll = -0.5 * resid @ np.linalg.solve(ky, resid) - 0.5 * np.linalg.slogdet(ky)[1] - 0.5 * N * np.log(2.0 * np.pi)
    np.testing.assert_allclose(out[0], want_ll, rtol=1e-9)
    assert out[5] == 0.0 and ogp is not None


@pytest.mark.parametrize("D,lat,N", [(5, 3, 7), (21, 3, 40), (51, 2, 9)])
def test_nested_sphere_projections_in_one_launch_match_the_level_by_level_path(D, lat, N):
    """projection_from_sphere_to_subsphere / projection_from_subsphere_to_sphere without an autograd graph (gabo_nested_sphere_project /
    _lift: every level in one launch) against the differentiable level-by-level statement of the same functions (pinned on the reference's
    values by test_gpu_parity's nested-sphere goldens): every entry of the returned lists."""
    from gabotorch_amd.nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere, projection_from_subsphere_to_sphere
    rng = np.random.default_rng(3 * D)
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    L = D - lat
    axes = []
    for k in range(L):
        a = rng.standard_normal(D - k)
        axes.append(T(a / np.linalg.norm(a)))
    dists = [torch.tensor([[v]], dtype=torch.float64) for v in rng.uniform(0.4, 2.6, L)]
    x = rng.standard_normal((N, D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    fused = projection_from_sphere_to_subsphere(T(x), axes, dists)
    stepwise = projection_from_sphere_to_subsphere(T(x, True), axes, dists)
    assert len(fused) == len(stepwise) == L + 1
    for a, b in zip(fused, stepwise):
        assert a.shape == b.shape
        np.testing.assert_allclose(a.cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-11, atol=1e-13)
    z = fused[-1].contiguous()
    up_fused = projection_from_subsphere_to_sphere(z, axes, dists)
    up_stepwise = projection_from_subsphere_to_sphere(z.clone().requires_grad_(True), axes, dists)
    assert len(up_fused) == len(up_stepwise) == L + 1
    for a, b in zip(up_fused, up_stepwise):
        assert a.shape == b.shape
        np.testing.assert_allclose(a.cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(np.linalg.norm(up_fused[-1].cpu().numpy(), axis=1), 1.0, atol=1e-4)      # (the reference's + 1e-6 in the normalisations)


@pytest.mark.parametrize("D,lat,n", [(5, 3, 12), (21, 3, 30), (51, 4, 20), (12, 2, 190)])
def test_nested_sphere_fit_objective_in_one_host_call_matches_autograd(D, lat, n):
    """fit_gpytorch_manifold's objective for ScaleKernel(NestedSphereGaussianKernel) - the surrogate of HD-GaBO on the sphere - as ONE host call
    (gabo_nested_sphere_fit_evaluate: projection through all levels, Gram, likelihood and the adjoints back to the axes) against the same
    marginal likelihood differentiated by torch autograd through the level-by-level statement (_MllProblem): value, the gradients w.r.t.
    every axis, beta, outputscale, noise and mean.  (The autograd gradients of the axes pass through their float32 parameters: 1e-6.)"""
    from gabotorch_amd.manifold_optimization import manifold_gp_fit as mgf
    from gabotorch_amd.manifold_optimization.host_manifolds import Euclidean, Product
    rng = np.random.default_rng(D + n)
    x = rng.standard_normal((n, D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    X = torch.tensor(x, device=DEV)
    y = torch.tensor(rng.standard_normal(n), device=DEV)
    torch.manual_seed(4)
    base = NestedSphereGaussianKernel(D, lat, beta_min=0.6)
    gp = models.SingleTaskGP(X, y, ScaleKernel(base, outputscale_prior=models.GammaPrior(2.0, 0.15)).double(), noise_prior=models.GammaPrior(1.1, 0.05))
    named = list(gp.named_parameters())
    params = [p for _, p in named]
    axis_ids = {id(a) for a in base.axes}
    factors = [getattr(base, nm.split(".")[-1] + "_manifold") if id(p) in axis_ids else Euclidean(int(p.numel())) for nm, p in named]
    manifold = Product(factors)
    fast = mgf._NestedSphereMllProblem.build(gp, [nm for nm, _ in named], params, manifold)
    assert fast is not None
    slow = mgf._MllProblem(gp, [nm for nm, _ in named], params, manifold)
    for trial in range(2):
        pt = [man.rand() if id(p) in axis_ids else rng.normal(0.3, 0.4, size=(int(p.numel()),)) for p, man in zip(params, factors)]
        fast.values_only = True
        c_values = fast.cost(pt)
        fast.values_only = False
        fast._recent = []
        c_fast, c_slow = fast.cost(pt), slow.cost(pt)
        np.testing.assert_allclose([c_values, c_fast], c_slow, rtol=1e-9)
        g_fast, g_slow = fast.egrad(pt), slow.egrad(pt)
        for a, b, (nm, p) in zip(g_fast, g_slow, named):
            tol = 2e-5 if id(p) in axis_ids else 1e-7
            np.testing.assert_allclose(np.ravel(a), np.ravel(b), rtol=tol, atol=tol * max(1e-3, float(np.abs(b).max())), err_msg=nm)
        # the flat-vector form the fit actually drives (PackedEuclideanSpheres): same numbers, no lists
        from gabotorch_amd.manifold_optimization.host_manifolds import PackedEuclideanSpheres
        packed = PackedEuclideanSpheres(factors)
        fast.flat_layout(packed._bounds)
        fast._recent = []
        v = packed.pack(pt)
        assert fast.cost_flat(v) == c_fast
        np.testing.assert_array_equal(fast.egrad_flat(v), packed.pack(g_fast))


def test_nested_spd_eigenvalue_constraints_golden(golden):
    """max/min_eigenvalue_nested_spd_constraint (bounds in the original space, hd_gabo_spd.py:245-256) with their gradients w.r.t.
    the nested point, single points and a batch, against the reference."""
    from gabotorch_amd.nested_mappings import nested_spd_constraints_utils as nscu
    g = golden("reconstruction.npz")
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    args = [T(g[f"a_{n}"]) for n in ("W", "V", "C", "K")]
    sym = lambda a: 0.5 * (a + a.T)     # noqa: E731
    for k in range(3):
        y = T(g["a_Y"][k], True)
        f = nscu.max_eigenvalue_nested_spd_constraint(y, 4.0, *args)
        np.testing.assert_allclose(float(f), g[f"nc_max{k}"], rtol=1e-10)
        f.backward()
        np.testing.assert_allclose(sym(y.grad.cpu().numpy()), sym(g[f"nc_gmax{k}"]), rtol=1e-7, atol=1e-9)
        y2 = T(g["a_Y"][k], True)
        f2 = nscu.min_eigenvalue_nested_spd_constraint(y2, 0.1, *args)
        np.testing.assert_allclose(float(f2), g[f"nc_min{k}"], rtol=1e-10)
        f2.backward()
        np.testing.assert_allclose(sym(y2.grad.cpu().numpy()), sym(g[f"nc_gmin{k}"]), rtol=1e-7, atol=1e-9)
    batch = nscu.max_eigenvalue_nested_spd_constraint(T(g["a_Y"][:3]), 4.0, *args)
    np.testing.assert_allclose(batch.cpu().numpy(), [g[f"nc_max{k}"] for k in range(3)], rtol=1e-10)
    sample = nscu.random_nested_spd_with_spd_eigenvalue_constraints(None, lambda: g["a_X"][0], args[0])
    np.testing.assert_allclose(sample, g["a_Y"][0], atol=1e-10)


# ---------------------------------------------------------------------------------------------- one-launch marginal likelihood
@pytest.mark.parametrize("n,power", [(1, 2), (2, 1), (5, 2), (37, 1), (64, 2), (65, 2), (128, 2), (160, 2)])
def test_gp_mll_kernel_matches_the_oracle(n, power):
    from gabotorch_amd import ops
    from oracle import gp as ogp
    rng = np.random.default_rng(n)
    pts = rng.standard_normal((n, 4))
    e = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1)) ** power
    y = rng.standard_normal(n)
    par = (0.4, 1.7, 0.03, -0.2)
    ll, g = ogp.marginal_log_likelihood(e, y, *par)
    out = ops.gp_mll(torch.tensor(e, device=DEV), torch.tensor(y, device=DEV), *par)
    assert out[5] == 0.0
    np.testing.assert_allclose(out[0], ll, rtol=1e-11)
    np.testing.assert_allclose(out[1:5], g, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(g).max()))


def test_gp_mll_kernel_reports_indefinite_matrices_and_sizes():
    from gabotorch_amd import _lib, ops
    e = torch.zeros(3, 3, dtype=torch.float64, device=DEV)       # K = os * ones + noise I with a negative "noise": indefinite
    out = ops.gp_mll(e, torch.ones(3, dtype=torch.float64, device=DEV), 1.0, 1.0, -0.5, 0.0)
    assert out[5] == 1.0 and out[:5] == [0.0] * 5
    n = _lib.GABO_GP_MLL_LARGE_MAX_N + 1
    with pytest.raises(RuntimeError):
        ops.gp_mll(torch.zeros(n, n, dtype=torch.float64, device=DEV), torch.zeros(n, dtype=torch.float64, device=DEV), 1.0, 1.0, 0.1, 0.0)
    n = _lib.GABO_GP_MLL_MAX_N + 40                                # the tiled path reports an indefinite matrix the same way
    out = ops.gp_mll(torch.zeros(n, n, dtype=torch.float64, device=DEV), torch.ones(n, dtype=torch.float64, device=DEV), 1.0, 1.0, -0.5, 0.0)
    assert out[5] == 1.0 and out[:5] == [0.0] * 5


@pytest.mark.parametrize("n", [161, 200, 256, 512, 1024])
def test_gp_mll_beyond_one_workgroup_against_the_oracle(n):
    """gabo_gp_mll_large (the sweep operator on 32 x 32 tiles, two launches per pivot block) against oracle/gp.py: likelihood, its four
    analytic derivatives, and W = alpha alpha^T - Ky^-1 of the Gram form."""
    from gabotorch_amd import ops
    from oracle import gp as ogp
    rng = np.random.default_rng(n)
    X = _rand_spd(rng, n, 3)
    e = ospd.affine_invariant_distance(X, X) ** 2
    y = rng.standard_normal(n)
    theta, os_, noise, mean = 0.8, 1.3, 0.05, 0.2
    want_ll, want_grad = ogp.marginal_log_likelihood(e, y, theta, os_, noise, mean)
    out = ops.gp_mll(torch.tensor(e, device=DEV), torch.tensor(y, device=DEV), theta, os_, noise, mean)
    assert out[5] == 0.0
    np.testing.assert_allclose(out[0], want_ll, rtol=1e-10)
    np.testing.assert_allclose(out[1:5], want_grad, rtol=1e-8, atol=1e-9 * abs(want_ll))
    kb = np.exp(-theta * e)
    o6, W = ops.gp_mll_gram(torch.tensor(kb, device=DEV), torch.tensor(y, device=DEV), os_, noise, mean)
    o6 = o6.tolist()
    np.testing.assert_allclose(o6[0], out[0], rtol=1e-12)
    np.testing.assert_allclose([o6[2], o6[3], o6[4]], [out[2], out[3], out[4]], rtol=1e-8)       # (exp(-theta e) formed on the host here, on the device there)
    ky = os_ * kb + noise * np.eye(n)
    alpha = np.linalg.solve(ky, y - mean)
    np.testing.assert_allclose(W.cpu().numpy(), np.outer(alpha, alpha) - np.linalg.inv(ky), rtol=1e-8, atol=1e-8 * np.abs(alpha).max() ** 2)


def _plain_models(rng, n=17):
    from gabotorch_amd.kernel_utils import kernels_sphere as ksph
    from gabotorch_amd.kernel_utils import kernels_spd as kspd
    X = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(_rand_spd(rng, n, 3)), device=DEV)
    S = rng.standard_normal((n, 4))
    S = torch.tensor(S / np.linalg.norm(S, axis=1, keepdims=True), device=DEV)
    y = torch.tensor(rng.standard_normal(n), device=DEV)
    prior = lambda: models.GammaPrior(2.0, 0.15)             # noqa: E731
    out = []
    for make, x in ((lambda: kspd.SpdAffineInvariantGaussianKernel(beta_min=0.3, beta_prior=models.GammaPrior(3.0, 2.0)), X),
                    (lambda: kspd.SpdAffineInvariantLaplaceKernel(beta_min=0.3), X),
                    (lambda: kspd.SpdFrobeniusGaussianKernel(), X),
                    (lambda: kspd.SpdLogEuclideanGaussianKernel(), X),
                    (lambda: ksph.SphereGaussianKernel(beta_min=0.5), S),
                    (lambda: ksph.SphereLaplaceKernel(), S)):
        out.append(lambda make=make, x=x: models.SingleTaskGP(x, y, ScaleKernel(make(), outputscale_prior=prior()),
                                                              noise_prior=models.GammaPrior(1.1, 0.05)))
    out.append(lambda: models.SingleTaskGP(S, y, ksph.SphereGaussianKernel(beta_min=0.5)))        # no ScaleKernel
    return out


def test_fast_mll_value_and_gradients_match_autograd_for_every_plain_kernel():
    for make in _plain_models(np.random.default_rng(3)):
        gp = make()
        with torch.no_grad():
            for p in gp.parameters():
                p.add_(0.3)
        ref = gp.marginal_log_likelihood()
        ref.backward()
        want = [p.grad.clone() for p in gp.parameters()]
        for p in gp.parameters():
            p.grad = None
        fast = gp._fast_mll_closure()
        assert fast is not None, type(gp.covar_module)
        value, proxy = fast()
        proxy.backward()
        np.testing.assert_allclose(value, float(ref), rtol=1e-11)
        for p, w in zip(gp.parameters(), want):
            np.testing.assert_allclose(p.grad.numpy(), w.numpy(), rtol=2e-6, atol=1e-9)     # raw_beta / lengthscale are fp32 parameters
        # the same through the plain-float chain rule that fit_gpytorch_model uses with the stand-in kernel classes
        params = list(gp.parameters())
        objective = gp._fast_scalar_objective(params)
        assert objective is not None
        loss, grad = objective(np.array([float(p) for p in params]))
        np.testing.assert_allclose(loss, -float(ref), rtol=1e-7)
        np.testing.assert_allclose(grad, [-float(w) for w in want], rtol=2e-6, atol=1e-9)


def test_fast_fit_reaches_the_same_hyper_parameters_as_the_autograd_fit():
    for make in _plain_models(np.random.default_rng(4))[:2] + _plain_models(np.random.default_rng(4))[4:5]:
        a, b = make(), make()
        models.fit_gpytorch_model(a, fast=True)
        models.fit_gpytorch_model(b, fast=False)
        np.testing.assert_allclose(float(a.marginal_log_likelihood()), float(b.marginal_log_likelihood()), rtol=1e-6, atol=1e-8)
        hyper = lambda m: [float(m.noise), float(m.mean_constant), float(m.covar_module.outputscale),      # noqa: E731
                           float(m.covar_module.base_kernel.beta)]
        np.testing.assert_allclose(hyper(a), hyper(b), rtol=2e-3, atol=1e-4)     # (raw parameters saturate where softplus is flat)


def test_nested_kernels_do_not_take_the_fast_fit():
    rng = np.random.default_rng(5)
    X = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(_rand_spd(rng, 6, 4)), device=DEV)
    gp = models.SingleTaskGP(X, torch.tensor(rng.standard_normal(6), device=DEV), ScaleKernel(NestedSpdLogEuclideanGaussianKernel(4, 2)))
    assert gp._fast_mll_closure() is None


def test_matrix_function_backward_from_saved_eigen_decomposition_matches_the_standalone_entry():
    """gabo_spd_matfun_backward_eig (what autograd uses: the forward launch's V, lambda) against gabo_spd_matfun_backward (solves the
    eigen-problem again) for log / exp / sqrt, incl. a repeated eigenvalue and the largest dimension."""
    from gabotorch_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for d in (1, 2, 5, 20, 32):
        a = _rand_spd(rng, 6, d)
        a[0] = 1.7 * np.eye(d)                                  # repeated eigenvalues: the divided differences must stay finite
        A = torch.tensor(a, device=DEV)
        G = torch.tensor(rng.standard_normal(a.shape), device=DEV)
        for op in (_lib.GABO_SPD_LOGM, _lib.GABO_SPD_EXPM, _lib.GABO_SPD_SQRTM):
            x = A.clone().requires_grad_(True)
            (ops.spd_matrix_function(x, op) * G).sum().backward()
            want = torch.empty_like(A)
            _lib.check(lib.gabo_spd_matfun_backward(op, A.data_ptr(), G.data_ptr(), want.data_ptr(), 6, d,
                                                    torch.cuda.current_stream().cuda_stream), "gabo_spd_matfun_backward")
            np.testing.assert_allclose(x.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-11, atol=1e-12)


def test_one_launch_likelihood_node_matches_the_torch_linalg_path():
    """marginal_log_likelihood (gabo_gp_mll_gram behind a custom autograd node) against the same quantity through torch.linalg's
    Cholesky / solve and autograd, for plain and nested kernels (parameters inside the distance), with and without ScaleKernel."""
    rng = np.random.default_rng(8)
    makers = _plain_models(rng)[:1] + _plain_models(rng)[3:5] + _plain_models(rng)[6:]
    X = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(_rand_spd(rng, 12, 4)), device=DEV)
    y = torch.tensor(rng.standard_normal(12), device=DEV)
    makers.append(lambda: models.SingleTaskGP(X, y, ScaleKernel(NestedSpdLogEuclideanGaussianKernel(4, 2)).double(),
                                              noise_prior=models.GammaPrior(1.1, 0.05)))
    for make in makers:
        gp = make()
        with torch.no_grad():
            for p in gp.parameters():
                if p.numel() == 1:
                    p.add_(0.2)
        a = gp.marginal_log_likelihood()
        a.backward()
        got = [None if p.grad is None else p.grad.clone() for p in gp.parameters()]
        for p in gp.parameters():
            p.grad = None
        b = gp._marginal_log_likelihood_torch()
        b.backward()
        np.testing.assert_allclose(float(a), float(b), rtol=1e-11)
        for p, w in zip(gp.parameters(), got):
            assert (w is None) == (p.grad is None)
            if w is not None:
                np.testing.assert_allclose(w.numpy(), p.grad.numpy(), rtol=2e-6, atol=1e-9)


def test_one_launch_likelihood_reports_indefinite_gram_matrices_as_minus_infinity():
    k = torch.tensor([[1.0, 2.0], [2.0, 1.0]], dtype=torch.float64, device=DEV, requires_grad=True)      # indefinite "kernel"
    one = torch.ones((), dtype=torch.float64)
    ll = models._ExactMll.apply(k, torch.zeros(2, dtype=torch.float64, device=DEV), one, 0.1 * one, 0.0 * one)
    assert float(ll) == -np.inf
    ll.backward()
    assert float(k.grad.abs().max()) == 0.0


def test_fused_reconstruction_launch_at_config5_size():
    """gabo_nested_spd_reconstruction at D = 20 -> d = 2 (config 5): values against the numpy oracle, gradients against the composition of
    the separate HIP launches it replaces (projection_from_nested_spd_to_spd + logm_torch / affine_invariant_distance_torch under autograd),
    a batch of parameter sets in one launch, and the host (numpy in / numpy out) form the optimiser uses."""
    from gabotorch_amd import _lib, ops
    from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd
    from gabotorch_amd.Riemannian_utils import spd_utils_torch as sut
    from oracle import spd as ospd
    rng = np.random.default_rng(21)
    _fused_reconstruction_case(rng, 20, 2, 13, 3)
    _fused_reconstruction_case(rng, 9, 3, 5, 2)              # complement of order 6: the padded order 8 of the wave eigen-solver; D = 9 -> 12


def _fused_reconstruction_case(rng, D, d, N, P):
    from gabotorch_amd import _lib, ops
    from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd
    from gabotorch_amd.Riemannian_utils import spd_utils_torch as sut
    from oracle import spd as ospd
    m = D - d
    X = _rand_spd(rng, N, D)
    Rm = np.linalg.qr(rng.standard_normal((D, D)))[0]
    W = Rm[:, :d]
    Y = np.einsum("ab,nac,cd->nbd", W, X, W)
    Vs = np.stack([np.linalg.qr(Rm[:, d:] + 0.05 * rng.standard_normal((D, m)))[0] for _ in range(P)])
    Cs = _rand_spd(rng, P, m)
    Ks = rng.standard_normal((P, d, m))
    Ks *= (0.3 + 0.5 * rng.uniform(size=(P, 1, 1))) / np.linalg.norm(Ks, axis=(1, 2), keepdims=True)
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    for metric, name in ((_lib.GABO_RECON_LOG_EUCLIDEAN, "le"), (_lib.GABO_RECON_AFFINE_INVARIANT, "ai")):
        rec = ops.NestedSpdReconstruction(T(X), T(Y), T(W), metric)
        costs, gV, gC, gK = rec.evaluate_host(Vs, Cs, Ks, grad=True)
        np.testing.assert_allclose(rec.evaluate_host(Vs, Cs, Ks, grad=False), costs, rtol=1e-14)
        for p in range(P):
            np.testing.assert_allclose(costs[p], ospd.reconstruction_cost(X, Y, W, Vs[p], Cs[p], Ks[p], metric=name), rtol=1e-10)
            V, C, K = T(Vs[p], True), T(Cs[p], True), T(Ks[p], True)
            xr = projection_from_nested_spd_to_spd(T(Y), T(W), V, C, K)
            if name == "le":
                diff = sut.logm_torch(T(X)) - sut.logm_torch(xr) + 1e-15
                ref = torch.sum(diff * diff)
            else:
                dist = sut.affine_invariant_distance_torch(T(X)[:, None], xr[:, None])
                ref = torch.sum(dist * dist)
            ref.backward()
            np.testing.assert_allclose(costs[p], ref.item(), rtol=1e-11)
            scale = max(float(V.grad.abs().max()), float(C.grad.abs().max()), float(K.grad.abs().max()))
            np.testing.assert_allclose(gV[p], V.grad.cpu().numpy(), rtol=1e-8, atol=1e-9 * scale)
            np.testing.assert_allclose(gC[p], C.grad.cpu().numpy(), rtol=1e-8, atol=1e-9 * scale)
            np.testing.assert_allclose(gK[p], K.grad.cpu().numpy(), rtol=1e-8, atol=1e-9 * scale)
            # single parameter set, host form and autograd form
            c1, v1, c1g, k1 = rec.evaluate_host(Vs[p], Cs[p], Ks[p], grad=True)
            np.testing.assert_allclose([c1], [costs[p]], rtol=1e-14)
            np.testing.assert_allclose(v1, gV[p], rtol=1e-13, atol=1e-13 * scale)
            V2, C2, K2 = T(Vs[p], True), T(Cs[p], True), T(Ks[p], True)
            (3.0 * rec(V2, C2, K2)).backward()
            np.testing.assert_allclose(K2.grad.cpu().numpy(), 3.0 * gK[p], rtol=1e-13, atol=1e-13 * scale)
            np.testing.assert_allclose(C2.grad.cpu().numpy(), 3.0 * gC[p], rtol=1e-13, atol=1e-13 * scale)


def test_nested_spd_eigenvalue_constraints_one_launch_at_config5_size():
    """gabo_nested_spd_extreme_eigenvalues (both extreme eigenpairs of the lifted point from one Householder reduction: multisection +
    inverse iteration) at D = 20 -> d = 2 and D = 12 -> d = 3, 6: values against numpy on the oracle's lift, gradients against the composed
    differentiable path (projection_from_nested_spd_to_spd + the eigenvalue op under autograd) that it replaces."""
    from gabotorch_amd.nested_mappings import nested_spd_constraints_utils as nscu
    from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd
    from gabotorch_amd.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch, min_eigenvalue_constraint_torch
    from oracle import spd as ospd
    rng = np.random.default_rng(5)
    T = lambda a, grad=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV, requires_grad=grad)   # noqa: E731
    for D, d, R in ((20, 2, 9), (12, 3, 4), (12, 6, 3), (7, 2, 5)):
        m = D - d
        Rm = np.linalg.qr(rng.standard_normal((D, D)))[0]
        W, V = Rm[:, :d], np.linalg.qr(Rm[:, d:] + 0.02 * rng.standard_normal((D, m)))[0]      # W^T V only approximately 0, as after the ALM
        C = _rand_spd(rng, 1, m)[0]
        K = rng.standard_normal((d, m))
        K *= 0.6 / np.linalg.norm(K)
        Y = _rand_spd(rng, R, d)
        args = [T(W), T(V), T(C), T(K)]
        lifted = ospd.projection_from_nested_spd_to_spd(Y, W, V, C, K)
        lam = np.linalg.eigvalsh(0.5 * (lifted + lifted.transpose(0, 2, 1)))
        y1, y2 = T(Y, True), T(Y, True)
        fmax = nscu.max_eigenvalue_nested_spd_constraint(y1, 5.0, *args)
        fmin = nscu.min_eigenvalue_nested_spd_constraint(y1, 0.1, *args)
        np.testing.assert_allclose(fmax.detach().cpu().numpy(), 5.0 - lam[:, -1], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(fmin.detach().cpu().numpy(), lam[:, 0] - 0.1, rtol=1e-12, atol=1e-13)
        wts = T(rng.standard_normal((2, R)))
        (wts[0] * fmax + wts[1] * fmin).sum().backward()
        xs = projection_from_nested_spd_to_spd(y2, *args)
        (wts[0] * max_eigenvalue_constraint_torch(xs, 5.0) + wts[1] * min_eigenvalue_constraint_torch(xs, 0.1)).sum().backward()
        g1, g2 = y1.grad.cpu().numpy(), y2.grad.cpu().numpy()
        np.testing.assert_allclose(0.5 * (g1 + g1.transpose(0, 2, 1)), 0.5 * (g2 + g2.transpose(0, 2, 1)), rtol=1e-9, atol=1e-11)
        # values only (the strict solver's feasibility test) and a single point
        with torch.no_grad():
            np.testing.assert_allclose(nscu.max_eigenvalue_nested_spd_constraint(T(Y[0]), 5.0, *args).item(), 5.0 - lam[0, -1], rtol=1e-12)


@pytest.mark.parametrize("flavour", ["log_euclidean", "affine_invariant"])
def test_nested_spd_fit_objective_without_autograd_matches_the_autograd_objective(flavour):
    """fit_gpytorch_manifold's objective for ScaleKernel(NestedSpd*GaussianKernel) - HD-GaBO's surrogate - as a chain of HIP launches
    with the backward launches called directly (_NestedSpdMllProblem) against the same marginal likelihood differentiated by torch
    autograd (_MllProblem): value and Euclidean gradients w.r.t. the projection matrix, lengthscale / beta, outputscale, noise and mean."""
    from gabotorch_amd.kernel_utils.kernels_spd import NestedSpdAffineInvariantGaussianKernel
    from gabotorch_amd.manifold_optimization import manifold_gp_fit as mgf
    from gabotorch_amd.manifold_optimization.host_manifolds import Euclidean, Product
    rng = np.random.default_rng(11)
    D, dl, n = 8, 2, 12
    X = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(_rand_spd(rng, n, D)), device=DEV)
    y = torch.tensor(rng.standard_normal(n), device=DEV)
    torch.manual_seed(2)
    base = NestedSpdLogEuclideanGaussianKernel(D, dl) if flavour == "log_euclidean" else NestedSpdAffineInvariantGaussianKernel(D, dl, beta_min=0.3)
    kern = ScaleKernel(base, outputscale_prior=models.GammaPrior(2.0, 0.15)).double()
    gp = models.SingleTaskGP(X, y, kern, noise_prior=models.GammaPrior(1.1, 0.05))
    named = list(gp.named_parameters())
    params = [p for _, p in named]
    factors = []
    for name, p in named:
        man = getattr(base, "raw_projection_matrix_manifold") if p is base.raw_projection_matrix else Euclidean(int(p.numel()))
        factors.append(man)
    manifold = Product(factors)
    fast = mgf._NestedSpdMllProblem.build(gp, [nm for nm, _ in named], params, manifold)
    assert fast is not None
    slow = mgf._MllProblem(gp, [nm for nm, _ in named], params, manifold)
    for trial in range(3):
        x = []
        for p, man in zip(params, factors):
            x.append(man.rand() if p is base.raw_projection_matrix else rng.normal(0.3, 0.5, size=(int(p.numel()),)))
        c_fast, c_slow = fast.cost(x), slow.cost(x)
        np.testing.assert_allclose(c_fast, c_slow, rtol=1e-10)
        g_fast, g_slow = fast.egrad(x), slow.egrad(x)
        for a, b, (nm, _) in zip(g_fast, g_slow, named):
            np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9, err_msg=nm)
        if flavour == "log_euclidean":
            # the same chain issued launch by launch from Python (what gabo_nested_spd_fit_evaluate replaces)
            assert fast.native and fast._recent
            fast.native = False
            c_chain, g_chain = fast.cost(x), fast.egrad(x)
            fast.native = True
            np.testing.assert_allclose(c_fast, c_chain, rtol=1e-13)
            for a, b, (nm, _) in zip(g_fast, g_chain, named):
                np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12, err_msg=nm)


@pytest.mark.parametrize("n", [40, 200])
def test_native_fit_evaluation_at_larger_training_sets(n):
    """gabo_nested_spd_fit_evaluate (one host call per evaluation of the HD-GaBO surrogate objective) against the launch-by-launch chain
    at training-set sizes on both sides of the one-workgroup likelihood limit (GABO_GP_MLL_MAX_N = 160), values only and with gradients."""
    from gabotorch_amd.manifold_optimization import manifold_gp_fit as mgf
    from gabotorch_amd.manifold_optimization.host_manifolds import Euclidean, Product
    rng = np.random.default_rng(n)
    D, dl = 10, 3
    X = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(_rand_spd(rng, n, D)), device=DEV)
    y = torch.tensor(rng.standard_normal(n), device=DEV)
    torch.manual_seed(3)
    base = NestedSpdLogEuclideanGaussianKernel(D, dl)
    gp = models.SingleTaskGP(X, y, ScaleKernel(base, outputscale_prior=models.GammaPrior(2.0, 0.15)).double(), noise_prior=models.GammaPrior(1.1, 0.05))
    named = list(gp.named_parameters())
    params = [p for _, p in named]
    factors = [getattr(base, "raw_projection_matrix_manifold") if p is base.raw_projection_matrix else Euclidean(int(p.numel())) for _, p in named]
    prob = mgf._NestedSpdMllProblem.build(gp, [nm for nm, _ in named], params, Product(factors))
    assert prob is not None and prob.native
    x = [man.rand() if p is base.raw_projection_matrix else rng.normal(0.3, 0.4, size=(int(p.numel()),)) for p, man in zip(params, factors)]
    prob.values_only = True
    c_values_only = prob.cost(x)
    prob.values_only = False
    prob._recent = []
    c_native, g_native = prob.cost(x), prob.egrad(x)
    prob.native = False
    c_chain, g_chain = prob.cost(x), prob.egrad(x)
    assert np.isfinite(c_chain)
    np.testing.assert_allclose([c_values_only, c_native], c_chain, rtol=1e-12)
    for a, b, (nm, _) in zip(g_native, g_chain, named):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11, err_msg=nm)


@pytest.mark.parametrize("n", [1, 7, 50, 64, 96])
def test_gp_factor_matches_lapack(n):
    """gabo_gp_factor (csrc/gp_factor.hip): L^-1, L^-T and alpha of outputscale K + noise I in one launch, against numpy's LAPACK factorisation
    of the same matrix - the quantities [3P] gpytorch caches for an ExactGP's posterior (behind manifold_optimize.py:182-184)."""
    import numpy as np
    import torch
    from gabotorch_amd import ops
    rng = np.random.default_rng(n)
    z = rng.standard_normal((n, 4))
    kb = np.exp(-0.7 * ((z[:, None] - z[None]) ** 2).sum(-1))
    y = rng.standard_normal(n)
    os_, noise, mean = 1.7, 1e-2, 0.3
    linv, linv_t, alpha, kinv = ops.gp_factor(torch.tensor(kb, device="cuda:0"), torch.tensor(y, device="cuda:0"), os_, noise, mean, want_kinv=True)
    K = os_ * kb + noise * np.eye(n)
    # the symmetric inverse the fused SPD acquisition kernels take in place of the two factors: exactly symmetric, = K^-1
    ki = kinv.cpu().numpy()
    np.testing.assert_array_equal(ki, ki.T)
    np.testing.assert_allclose(ki, np.linalg.inv(K), rtol=0, atol=1e-9 * np.abs(ki).max())
    L = np.linalg.cholesky(K)
    want = np.linalg.inv(L)
    scale = np.abs(want).max()
    np.testing.assert_allclose(linv.cpu().numpy(), want, rtol=0, atol=1e-10 * scale)
    np.testing.assert_array_equal(linv_t.cpu().numpy(), linv.cpu().numpy().T)
    assert np.all(np.triu(linv.cpu().numpy(), 1) == 0.0)          # exact zeros above the diagonal (the SPD kernels sum whole rows)
    a_want = np.linalg.solve(K, y - mean)
    np.testing.assert_allclose(alpha.cpu().numpy(), a_want, rtol=0, atol=1e-9 * np.abs(a_want).max())


def test_gp_factor_refuses_what_it_cannot_factor():
    import numpy as np
    import torch
    from gabotorch_amd import _lib, ops
    bad = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(RuntimeError, match="not positive definite"):
        ops.gp_factor(torch.tensor(bad, device="cuda:0"), torch.zeros(2, dtype=torch.float64, device="cuda:0"), 1.0, 0.0, 0.0)
    ops.gp_factor(torch.eye(3, dtype=torch.float64, device="cuda:0"), torch.zeros(3, dtype=torch.float64, device="cuda:0"), 1.0, 0.0, 0.0)   # the status word is clean again
    n = _lib.GABO_GP_FACTOR_MAX_N + 1
    with pytest.raises(RuntimeError):
        ops.gp_factor(torch.eye(n, dtype=torch.float64, device="cuda:0"), torch.zeros(n, dtype=torch.float64, device="cuda:0"), 1.0, 0.0, 0.0)


def test_exact_gp_prediction_cache_through_gp_factor_equals_the_torch_route():
    """models.ExactGP / SingleTaskGP fill their prediction cache with ONE launch on the device; the posterior is the one the torch route gives"""
    import numpy as np
    import torch
    from gabotorch_amd import models
    from gabotorch_amd._compat import ScaleKernel
    from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
    from oracle import spd as ospd
    rng = np.random.default_rng(5)
    q = np.linalg.qr(rng.standard_normal((30, 3, 3)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.3, 3.0, (30, 3)), q)
    xv = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(0.5 * (X + X.transpose(0, 2, 1))), device="cuda:0")
    y = torch.tensor(rng.standard_normal(30), device="cuda:0")
    k = SpdAffineInvariantGaussianKernel(beta_min=0.25)
    gp = models.ExactGP(xv, y, k, outputscale=1.3, noise=1e-2)
    linv, alpha = gp._train_cache()
    assert getattr(gp, "_cache_linv_t", None) is not None          # the one-launch route ran
    kb = k.forward(xv, xv).detach().cpu().numpy()
    K = 1.3 * kb + 1e-2 * np.eye(30)
    np.testing.assert_allclose(linv.cpu().numpy(), np.linalg.inv(np.linalg.cholesky(K)), rtol=0, atol=1e-9 * np.abs(linv.cpu().numpy()).max())
    np.testing.assert_allclose(alpha.cpu().numpy(), np.linalg.solve(K, y.cpu().numpy() - gp.mean), rtol=1e-8, atol=1e-10)
    st = models.SingleTaskGP(xv, y, ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=0.25)))
    li2, al2, mu2 = st._ensure_cache()
    assert getattr(st, "_cache_linv_t", None) is not None
    K2 = st._kxx().detach().cpu().numpy()
    np.testing.assert_allclose(al2.cpu().numpy(), np.linalg.solve(K2, y.cpu().numpy() - float(mu2)), rtol=1e-8, atol=1e-10)
