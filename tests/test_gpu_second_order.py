"""Second-order autograd through the SPD affine-invariant kernels (gabo_spd_ai_backward2): Hessian-vector products with respect to x1, as the
reference's PyTorch backend builds them for exact Hessians (pymanopt_addons/tools/autodiff/_pytorch.py:103-116: torch.autograd.grad of
<gradient, vector> through cholesky / inverse / bmm / symeig(eigenvectors=True) / log / exp, Riemannian_utils/spd_utils_torch.py:87-120).
The checker is the same op sequence in torch on the CPU in float64, differentiated twice by torch itself."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, ops
from oracle import spd as ospd
from oracle import sphere as osph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = {"gaussian": _lib.GABO_OUT_GAUSSIAN, "laplace": _lib.GABO_OUT_LAPLACE, "distance": _lib.GABO_OUT_DISTANCE}


def _rand_spd_mandel(rng, n, d, lo=0.3, hi=3.0):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return ospd.symmetric_matrix_to_vector_mandel(0.5 * (m + m.transpose(0, 2, 1)))


def _mandel_operator(d):
    """(d*d) x d_vec matrix P with vec(matrix) = P v  (vector_to_symmetric_matrix_mandel, spd_utils_torch.py:159-194)"""
    dv = d * (d + 1) // 2
    P = np.zeros((d * d, dv))
    for e in range(dv):
        v = np.zeros(dv)
        v[e] = 1.0
        P[:, e] = ospd.vector_to_symmetric_matrix_mandel(v[None])[0].reshape(-1)
    return torch.tensor(P)


def _torch_kernel(x1v, x2v, beta, mode, d):
    """the reference's chain (spd_utils_torch.py:87-120, kernels_spd.py:94-98, 185) in plain torch, float64, CPU"""
    P = _mandel_operator(d)
    A = (x1v @ P.T).reshape(-1, d, d)
    B = (x2v @ P.T).reshape(-1, d, d)
    Linv = torch.linalg.inv(torch.linalg.cholesky(A))
    M = Linv[:, None] @ B[None] @ Linv[:, None].transpose(-1, -2)
    lam = torch.linalg.eigvalsh(0.5 * (M + M.transpose(-1, -2)))
    dist = torch.sqrt((lam.log() ** 2).sum(-1) + 1e-15)
    if mode == "distance":
        return dist
    return torch.exp(-beta * dist) if mode == "laplace" else torch.exp(-beta * dist ** 2)


@pytest.mark.parametrize("d", [2, 3, 5, 8, 10])
@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
def test_hessian_vector_product_matches_torch_double_backward(d, mode):
    rng = np.random.default_rng(100 + d)
    n1, n2, beta = 6, 9, 0.7
    x1, x2 = _rand_spd_mandel(rng, n1, d), _rand_spd_mandel(rng, n2, d)
    G = rng.standard_normal((n1, n2))
    U = rng.standard_normal(x1.shape)
    # torch on the CPU: g = d sum(G K) / d x1 with its graph, then d <g, U> / d (x1, G)
    a = torch.tensor(x1, requires_grad=True)
    Gc = torch.tensor(G, requires_grad=True)
    Kc = _torch_kernel(a, torch.tensor(x2), beta, mode, d)
    (gc,) = torch.autograd.grad((Gc * Kc).sum(), a, create_graph=True)
    hv_c, dg_c = torch.autograd.grad((gc * torch.tensor(U)).sum(), (a, Gc))
    # the HIP path through the same two autograd calls
    ad = torch.tensor(x1, device=DEV, requires_grad=True)
    Gd = torch.tensor(G, device=DEV, requires_grad=True)
    Kd = ops.spd_ai_kernel(ad, torch.tensor(x2, device=DEV), beta, MODES[mode])
    (gd,) = torch.autograd.grad((Gd * Kd).sum(), ad, create_graph=True)
    np.testing.assert_allclose(gd.detach().cpu().numpy(), gc.detach().numpy(), rtol=1e-9, atol=1e-11)
    hv_d, dg_d = torch.autograd.grad((gd * torch.tensor(U, device=DEV)).sum(), (ad, Gd))
    scale = float(hv_c.abs().max())
    np.testing.assert_allclose(hv_d.cpu().numpy(), hv_c.numpy(), rtol=1e-8, atol=1e-9 * scale)
    np.testing.assert_allclose(dg_d.cpu().numpy(), dg_c.numpy(), rtol=1e-8, atol=1e-10 * float(dg_c.abs().max()))
    # the Hessian of sum(G K) is symmetric: <V, H U> = <U, H V>
    V = torch.tensor(rng.standard_normal(x1.shape), device=DEV)
    hv_v = ops.spd_ai_backward2(ad.detach(), torch.tensor(x2, device=DEV), Gd.detach(), V, beta, MODES[mode])[0]
    lhs = float((V * hv_d).sum())
    rhs = float((torch.tensor(U, device=DEV) * hv_v).sum())
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), scale)


@pytest.mark.parametrize("d", [2, 3, 5])
@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
def test_full_hessian_in_both_arguments_matches_torch_double_backward(d, mode):
    """both arguments carry gradients: diagonal blocks, mixed blocks and the x2 side (the same kernel with the sets exchanged)"""
    rng = np.random.default_rng(200 + d)
    n1, n2, beta = 5, 7, 0.6
    x1, x2 = _rand_spd_mandel(rng, n1, d), _rand_spd_mandel(rng, n2, d)
    G = rng.standard_normal((n1, n2))
    U1, U2 = rng.standard_normal(x1.shape), rng.standard_normal(x2.shape)

    def run(dev, kernel):
        a = torch.tensor(x1, device=dev, requires_grad=True)
        b = torch.tensor(x2, device=dev, requires_grad=True)
        K = kernel(a, b)
        g1, g2 = torch.autograd.grad((torch.tensor(G, device=dev) * K).sum(), (a, b), create_graph=True)
        h1, h2 = torch.autograd.grad((g1 * torch.tensor(U1, device=dev)).sum() + (g2 * torch.tensor(U2, device=dev)).sum(), (a, b))
        return [t_.detach().cpu().numpy() for t_ in (g1, g2, h1, h2)]
    want = run("cpu", lambda a, b: _torch_kernel(a, b, beta, mode, d))
    got = run(DEV, lambda a, b: ops.spd_ai_kernel(a, b, beta, MODES[mode]))
    for w, g in zip(want, got):
        np.testing.assert_allclose(g, w, rtol=1e-8, atol=1e-9 * np.abs(w).max())


def test_kernel_of_a_point_with_itself_has_no_curvature():
    """k(x, x) is constant: the two diagonal blocks of its Hessian are cancelled by the mixed ones (what a posterior variance differentiates)"""
    rng = np.random.default_rng(9)
    x = torch.tensor(_rand_spd_mandel(rng, 4, 3)[:, None], device=DEV, requires_grad=True)      # batch of 4 single points
    K = ops.spd_ai_kernel(x, x, 0.8, _lib.GABO_OUT_GAUSSIAN)
    (g,) = torch.autograd.grad(K.sum(), x, create_graph=True)
    (h,) = torch.autograd.grad((g * torch.tensor(rng.standard_normal(tuple(x.shape)), device=DEV)).sum(), x)
    assert float(g.detach().abs().max()) < 1e-6 and float(h.abs().max()) < 1e-6       # (d(x, x) = sqrt(1e-15): the reference's own derivative is noise of this size)


def test_hessian_vector_product_with_repeated_and_close_eigenvalues():
    """x2 = a multiple of x1 (M = c I: every divided difference is a derivative) and pairs with nearly equal eigenvalues of M"""
    rng = np.random.default_rng(5)
    d = 4
    x1 = _rand_spd_mandel(rng, 3, d)
    m1 = ospd.vector_to_symmetric_matrix_mandel(x1)
    x2m = np.stack([2.5 * m1[0], m1[1] * (1.0 + 1e-9) + 1e-10 * np.diag(np.arange(d)), m1[2] @ m1[2] / 2.0, 0.7 * m1[0]])
    x2 = ospd.symmetric_matrix_to_vector_mandel(0.5 * (x2m + x2m.transpose(0, 2, 1)))
    G = rng.standard_normal((3, 4))
    U = rng.standard_normal(x1.shape)
    hv, dg, _ = ops.spd_ai_backward2(torch.tensor(x1, device=DEV), torch.tensor(x2, device=DEV), torch.tensor(G, device=DEV), torch.tensor(U, device=DEV), 0.9)
    # finite differences of the first-order HIP gradient along U (central, h = 1e-5: error O(h^2) ~ 1e-9 of the third derivative)
    h = 1e-5
    gp = ops.spd_ai_backward(torch.tensor(x1 + h * U, device=DEV), torch.tensor(x2, device=DEV), torch.tensor(G, device=DEV), 0.9)
    gm = ops.spd_ai_backward(torch.tensor(x1 - h * U, device=DEV), torch.tensor(x2, device=DEV), torch.tensor(G, device=DEV), 0.9)
    fd = ((gp - gm) / (2 * h)).cpu().numpy()
    np.testing.assert_allclose(hv.cpu().numpy(), fd, rtol=2e-6, atol=2e-6 * np.abs(fd).max())
    kp = ops.spd_ai_pairwise(torch.tensor(x1 + h * U, device=DEV), torch.tensor(x2, device=DEV), beta=0.9)
    km = ops.spd_ai_pairwise(torch.tensor(x1 - h * U, device=DEV), torch.tensor(x2, device=DEV), beta=0.9)
    np.testing.assert_allclose(dg.cpu().numpy(), ((kp - km) / (2 * h)).cpu().numpy(), rtol=2e-6, atol=1e-8)


def test_exact_hessian_of_an_acquisition_on_the_spd_manifold():
    """approx_hessian=False on S^d_++ (reference: problem.hess = ehess2rhess(egrad, ehess) with ehess from the second autograd pass,
    pymanopt_addons/problem.py:143-159): the batched problem's exact Riemannian Hessian against central differences of its own Euclidean
    gradient, and the maximiser reaches the optimum of the finite-difference-Hessian run."""
    from gabotorch_amd import manifolds, models
    from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedProblem, BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch
    rng = np.random.default_rng(0)
    d = 3
    xv = torch.tensor(_rand_spd_mandel(rng, 12, d, 0.5, 2.0), device=DEV)
    y = torch.tensor(rng.standard_normal(12), device=DEV)
    gp = models.ExactGP(xv, y, SpdAffineInvariantGaussianKernel(beta_min=0.5), outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    man = manifolds.PositiveDefinite(d)

    def cost(x):                                     # x: R x d x d matrices -> R values (manifold_optimize.py:175-184)
        return -acq(symmetric_matrix_to_vector_mandel_torch(x)[:, None])
    x0 = vector_to_symmetric_matrix_mandel_torch(torch.tensor(_rand_spd_mandel(rng, 4, d, 0.6, 1.8), device=DEV))
    u = torch.tensor(rng.standard_normal((4, d, d)), device=DEV)
    u = 0.5 * (u + u.transpose(1, 2))
    P = BatchedProblem(man, cost, approx_hessian=False)
    exact = P.hess(x0, u)
    # the Euclidean Hessian-vector product by central differences of the Euclidean gradient (x0 +- h u stays symmetric positive definite),
    # carried to the manifold by the same ehess2rhess
    h = 1e-5
    eh_fd = (P.cost_egrad(x0 + h * u)[1] - P.cost_egrad(x0 - h * u)[1]) / (2 * h)
    want = man.ehess2rhess(x0, P.cost_egrad(x0)[1], eh_fd, u)
    np.testing.assert_allclose(exact.cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-6 * float(want.abs().max()))
    starts = symmetric_matrix_to_vector_mandel_torch(x0)[:, None]
    best = {}
    for approx in (False, True):
        cand, val = gen_candidates_manifold(starts, acq, man, BatchedTrustRegions(maxiter=60), vector_to_symmetric_matrix_mandel_torch,
                                            symmetric_matrix_to_vector_mandel_torch, approx_hessian=approx)
        best[approx] = float(val.max())
    assert abs(best[False] - best[True]) <= 1e-5 * max(abs(best[True]), 1e-3), best


@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
def test_second_derivatives_involving_beta_match_torch_double_backward(mode):
    """d^2/dbeta^2, d^2/dbeta dx (from the beta gradient) and d/dbeta <dK/dx, U> (from the x gradient) under create_graph=True -
    the marginal likelihood's Hessian with respect to beta and the x-beta blocks - complete, not the partial value of a constant saved matrix"""
    rng = np.random.default_rng(31)
    d, n1, n2 = 3, 5, 6
    x1, x2 = _rand_spd_mandel(rng, n1, d), _rand_spd_mandel(rng, n2, d)
    G, U1, U2 = rng.standard_normal((n1, n2)), rng.standard_normal(x1.shape), rng.standard_normal(x2.shape)

    def run(dev, kernel):
        a = torch.tensor(x1, device=dev, requires_grad=True)
        b = torch.tensor(x2, device=dev, requires_grad=True)
        beta = torch.tensor(0.7, dtype=torch.float64, requires_grad=True)
        Gt = torch.tensor(G, device=dev, requires_grad=True)
        L = (Gt * kernel(a, b, beta)).sum()
        gb, ga, gbv = torch.autograd.grad(L, (beta, a, b), create_graph=True, allow_unused=True)
        outs = []
        if mode == "distance":          # beta does not enter: the gradient is zero (or absent), its derivatives too
            assert gb is None or float(gb.detach().abs()) == 0.0
            gb = None
        if gb is not None:
            outs += list(torch.autograd.grad(gb, (beta, a, b, Gt), retain_graph=True))
        mixed = (ga * torch.tensor(U1, device=dev)).sum() + (gbv * torch.tensor(U2, device=dev)).sum()
        (mb,) = torch.autograd.grad(mixed, beta, allow_unused=True)
        outs.append(torch.zeros(()) if mb is None else mb)
        return [o.detach().cpu().numpy() for o in outs]
    want = run("cpu", lambda a, b, beta: _torch_kernel(a, b, beta, mode, d))
    got = run(DEV, lambda a, b, beta: ops.spd_ai_kernel(a, b, beta, MODES[mode]))
    assert len(want) == len(got)
    for w, g in zip(want, got):
        np.testing.assert_allclose(g, w, rtol=1e-8, atol=1e-9 * max(np.abs(w).max(), 1e-3))


@pytest.mark.parametrize("which", ["x2", "x1"])
def test_gradients_of_an_expanded_set_are_not_counted_twice(which):
    """a training set handed over as .expand(b, n, dv) (gpytorch's train inputs in a batched acquisition call, batch stride 0):
    first- and second-order gradients with respect to the BASE tensor against torch on the CPU"""
    rng = np.random.default_rng(32)
    d, nb, n1, n2 = 3, 4, 3, 5
    base = _rand_spd_mandel(rng, n2, d)
    other = np.stack([_rand_spd_mandel(rng, n1, d) for _ in range(nb)])
    G, U = rng.standard_normal((nb, n1, n2)), rng.standard_normal(other.shape)

    def run(dev, kernel):
        bt = torch.tensor(base, device=dev, requires_grad=True)
        ot = torch.tensor(other, device=dev, requires_grad=True)
        ex = bt.expand(nb, n2, base.shape[-1])
        K = kernel(ot, ex) if which == "x2" else kernel(ex, ot).transpose(-1, -2)
        L = (torch.tensor(G, device=dev) * K).sum()
        g_base, g_other = torch.autograd.grad(L, (bt, ot), create_graph=True)
        (h_base,) = torch.autograd.grad((g_other * torch.tensor(U, device=dev)).sum(), bt)
        return [t_.detach().cpu().numpy() for t_ in (g_base, g_other, h_base)]

    def torch_kernel(a, b):
        return torch.stack([_torch_kernel(a[k], b[k], 0.8, "gaussian", d) for k in range(nb)])
    want = run("cpu", torch_kernel)
    got = run(DEV, lambda a, b: ops.spd_ai_kernel(a, b, 0.8, _lib.GABO_OUT_GAUSSIAN))
    for w, g in zip(want, got):
        np.testing.assert_allclose(g, w, rtol=1e-8, atol=1e-9 * np.abs(w).max())


# ---- pinned to the reference: `ehess` of its own PytorchBackend (tests/golden/make_golden_hvp.py -> hvp.npz) ----------------------------------
def _mandel(m):
    return ospd.symmetric_matrix_to_vector_mandel(0.5 * (m + np.swapaxes(m, -1, -2)))


@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
@pytest.mark.parametrize("d", [2, 3, 5, 10])
def test_hessian_vector_product_against_the_reference_backend(golden, d, mode):
    """gabo_spd_ai_backward2 (directly and through two autograd passes) against the Hessian-vector products the reference's PyTorch backend
    returns (pymanopt_addons/tools/autodiff/_pytorch.py:103-116) and against the oracle's closed form; the set includes a pair whose M has a
    spectrum repeated to 1e-4"""
    z = golden("hvp.npz")
    x1, x2, G, U, beta = (z[f"spd{d}_{k}"] for k in ("x1", "x2", "G", "U", "beta"))
    want = _mandel(z[f"spd{d}_{mode}_ehess"])
    want_g = _mandel(z[f"spd{d}_{mode}_egrad"])
    t = lambda a: torch.tensor(a, device=DEV)       # noqa: E731
    hv = ops.spd_ai_backward2(t(_mandel(x1)), t(_mandel(x2)), t(G), t(_mandel(U)), float(beta), MODES[mode])[0].cpu().numpy()
    np.testing.assert_allclose(hv, want, rtol=0, atol=1e-9 * np.abs(want).max())
    np.testing.assert_allclose(hv, ospd.spd_ai_kernel_hvp(_mandel(x1), _mandel(x2), float(beta), G, _mandel(U), mode), rtol=0, atol=1e-9 * np.abs(want).max())
    a = t(_mandel(x1)).requires_grad_()
    K = ops.spd_ai_kernel(a, t(_mandel(x2)), float(beta), MODES[mode])
    (g,) = torch.autograd.grad((t(G) * K).sum(), a, create_graph=True)
    np.testing.assert_allclose(g.detach().cpu().numpy(), want_g, rtol=0, atol=1e-9 * np.abs(want_g).max())
    (h,) = torch.autograd.grad((g * t(_mandel(U))).sum(), a)
    np.testing.assert_allclose(h.cpu().numpy(), want, rtol=0, atol=1e-9 * np.abs(want).max())


@pytest.mark.parametrize("dim", [3, 5, 10])
def test_sphere_hessian_vector_product_against_the_reference_backend(golden, dim):
    """the sphere kernel's second autograd pass (gabo_sphere_from_inner, order 2) against the reference backend's `ehess` and the oracle"""
    z = golden("hvp.npz")
    x1, x2, G, U, beta = (z[f"sph{dim}_{k}"] for k in ("x1", "x2", "G", "U", "beta"))
    t = lambda a: torch.tensor(a, device=DEV)       # noqa: E731
    a = t(x1).requires_grad_()
    K = ops.sphere_kernel(a, t(x2), float(beta), _lib.GABO_OUT_GAUSSIAN)
    (g,) = torch.autograd.grad((t(G) * K).sum(), a, create_graph=True)
    np.testing.assert_allclose(g.detach().cpu().numpy(), z[f"sph{dim}_egrad"], rtol=0, atol=1e-10 * np.abs(z[f"sph{dim}_egrad"]).max())
    (h,) = torch.autograd.grad((g * t(U)).sum(), a)
    want = z[f"sph{dim}_ehess"]
    np.testing.assert_allclose(h.cpu().numpy(), want, rtol=0, atol=1e-10 * np.abs(want).max())
    np.testing.assert_allclose(osph.sphere_gaussian_kernel_hvp(x1, x2, float(beta), G, U), want, rtol=0, atol=1e-12 * np.abs(want).max())
