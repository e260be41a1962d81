"""Gradients of the HIP kernels (closed-form backward kernels, autograd plumbing, kernel classes) against the golden
autograd vectors of the reference and against the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel, SpdAffineInvariantLaplaceKernel
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel, SphereLaplaceKernel
from gabotorch_amd.Riemannian_utils import spd_utils_torch as sut
from gabotorch_amd.Riemannian_utils import sphere_utils_torch as sphut
from oracle import spd as ospd
from oracle import sphere as osph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x, grad=False):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV, requires_grad=grad)


def rand_spd_mandel(rng, n, d, lo=0.05, hi=5.0):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return ospd.symmetric_matrix_to_vector_mandel(0.5 * (m + m.transpose(0, 2, 1)))


def test_spd_grads_golden(golden):
    g = golden("spd_ai.npz")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        x1, x2 = t(g[p + "x1"], True), t(g[p + "x2"], True)
        k = ops.spd_ai_kernel(x1, x2, float(g[p + "beta"]))
        (k * t(g[p + "gup"])).sum().backward()
        for mine, ref in ((x1.grad, g[p + "grad_x1"]), (x2.grad, g[p + "grad_x2"])):
            # the reference gradient carries its fp32 eigenvalue sink (spd_utils_torch.py:108): ~1e-6 of the row scale
            np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(ref).max()))
        o1, o2 = ospd.spd_ai_gaussian_kernel_grads(g[p + "x1"], g[p + "x2"], float(g[p + "beta"]), g[p + "gup"])
        np.testing.assert_allclose(x1.grad.cpu().numpy(), o1, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(o1).max()))
        np.testing.assert_allclose(x2.grad.cpu().numpy(), o2, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(o2).max()))


@pytest.mark.parametrize("d", list(range(2, _lib.GABO_SPD_BWD_REG_MAX_DIM + 1)) + [17, 20, 24, 32])
def test_spd_backward_all_dims_vs_oracle(d):
    """d <= 11: one lane per pair; 12 ... 16: two lanes per pair (odd d: a padded half row in the odd lane; 70 and 33 columns: a ragged last chunk of the
    32-pair waves); above: the wave-per-pair fallback"""
    rng = np.random.default_rng(40 + d)
    x1, x2 = rand_spd_mandel(rng, 5, d), rand_spd_mandel(rng, 70 if d % 3 else 33, d)
    gup = rng.standard_normal((5, x2.shape[0]))
    got = ops.spd_ai_backward(t(x1), t(x2), t(gup), 0.8, _lib.GABO_OUT_GAUSSIAN, wrt=1).cpu().numpy()
    want, want2 = ospd.spd_ai_gaussian_kernel_grads(x1, x2, 0.8, gup)
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10 * np.abs(want).max())
    got2 = ops.spd_ai_backward(t(x1), t(x2), t(gup), 0.8, _lib.GABO_OUT_GAUSSIAN, wrt=2).cpu().numpy()
    np.testing.assert_allclose(got2, want2, rtol=1e-8, atol=1e-10 * np.abs(want2).max())


def test_spd_distance_and_laplace_grads_by_finite_differences():
    rng = np.random.default_rng(3)
    d = 4
    x1, x2 = rand_spd_mandel(rng, 3, d, 0.3, 3.0), rand_spd_mandel(rng, 6, d, 0.3, 3.0)
    gup = rng.standard_normal((3, 6))
    for mode, fn in ((_lib.GABO_OUT_DISTANCE, lambda a: ospd.affine_invariant_distance(ospd.vector_to_symmetric_matrix_mandel(a),
                                                                                        ospd.vector_to_symmetric_matrix_mandel(x2))),
                     (_lib.GABO_OUT_LAPLACE, lambda a: ospd.spd_ai_laplace_kernel(a, x2, 0.7))):
        got = ops.spd_ai_backward(t(x1), t(x2), t(gup), 0.7, mode, wrt=1).cpu().numpy()
        fd = np.zeros_like(x1)
        h = 1e-6
        for i in range(x1.shape[0]):
            for e in range(x1.shape[1]):
                xp, xm = x1.copy(), x1.copy()
                xp[i, e] += h
                xm[i, e] -= h
                fd[i, e] = ((fn(xp) - fn(xm)) * gup).sum() / (2 * h)
        np.testing.assert_allclose(got, fd, rtol=2e-6, atol=1e-7)


def test_spd_kernel_classes_and_beta_gradient(golden):
    g = golden("spd_ai.npz")
    x1, x2, beta = g["c3_x1"], g["c3_x2"], float(g["c3_beta"])      # d = 10
    k = SpdAffineInvariantGaussianKernel(beta_min=0.0)
    k.beta = beta
    out = k.forward(t(x1), t(x2))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["c3_K"], rtol=1e-5, atol=1e-7)
    assert out.dtype == torch.float64 and out.shape == (12, 17)
    # d/d raw_beta through softplus: compare with the oracle by finite differences on beta
    gup = t(g["c3_gup"])
    (out * gup).sum().backward()
    bb = k.beta.item()
    h = 1e-6
    fd = ((ospd.spd_ai_gaussian_kernel(x1, x2, bb + h) - ospd.spd_ai_gaussian_kernel(x1, x2, bb - h)) * g["c3_gup"]).sum() / (2 * h)
    dbeta_draw = torch.sigmoid(k.raw_beta.detach()).item()           # d softplus
    np.testing.assert_allclose(k.raw_beta.grad.item(), fd * dbeta_draw, rtol=1e-5)
    # diagonal_distance contract and the Laplace twin
    ones = k.forward(t(x1), t(x2), diagonal_distance=True)
    assert ones.shape == (17, 1) and torch.all(ones == 1)
    kl = SpdAffineInvariantLaplaceKernel(beta_min=0.1)
    np.testing.assert_allclose(kl.forward(t(x1), t(x2)).detach().cpu().numpy(),
                               ospd.spd_ai_laplace_kernel(x1, x2, kl.beta.item()), rtol=1e-9)
    # acquisition-shaped call: candidates (b,1,dv) against an expand()ed training set, gradient w.r.t. the candidates
    rng = np.random.default_rng(9)
    train = rand_spd_mandel(rng, 13, 5)
    cand = np.stack([rand_spd_mandel(rng, 1, 5) for _ in range(20)])
    kk = SpdAffineInvariantGaussianKernel(beta_min=0.25)
    xc = t(cand, True)
    res = kk.forward(xc, t(train).expand(20, 13, 15))
    w = rng.standard_normal((20, 1, 13))
    (res * t(w)).sum().backward()
    want, _ = ospd.spd_ai_gaussian_kernel_grads(cand, np.broadcast_to(train, (20, 13, 15)), kk.beta.item(), w)
    np.testing.assert_allclose(xc.grad.cpu().numpy(), want, rtol=1e-8, atol=1e-11)


def test_affine_invariant_distance_torch_matrix_interface(golden):
    g = golden("spd_ai.npz")
    m1 = ospd.vector_to_symmetric_matrix_mandel(g["c2_x1"])
    m2 = ospd.vector_to_symmetric_matrix_mandel(g["c2_x2"])
    a = t(m1, True)
    dist = sut.affine_invariant_distance_torch(a, t(m2))
    np.testing.assert_allclose(dist.detach().cpu().numpy(), g["c2_dist"], rtol=5e-7, atol=5e-7)
    dist.sum().backward()
    # symmetric gradient w.r.t. the matrix entries (both triangles), as the reference's autograd gives
    ga = a.grad.cpu().numpy()
    np.testing.assert_allclose(ga, ga.transpose(0, 2, 1), rtol=1e-12, atol=1e-14)
    z = sut.affine_invariant_distance_torch(t(np.zeros((3, 4, 2, 2))), t(np.zeros((3, 6, 2, 2))), diagonal_distance=True)
    assert tuple(z.shape) == tuple(g["diag_shape"])
    v = t(g["c2_x1"], True)
    mm = sut.vector_to_symmetric_matrix_mandel_torch(v)
    (mm * mm).sum().backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), 2 * g["c2_x1"], rtol=1e-12)      # ||M||_F^2 = ||v||^2 in Mandel form


def test_sphere_grads_golden_and_classes(golden):
    g = golden("sphere.npz")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        x1, x2 = t(g[p + "x1"], True), t(g[p + "x2"], True)
        k = SphereGaussianKernel(beta_min=0.0)
        k.beta = float(g[p + "beta"])
        out = ops.sphere_kernel(x1, x2, float(g[p + "beta"]))
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[p + "K"], rtol=1e-11, atol=1e-14)
        (out * t(g[p + "gup"])).sum().backward()
        for mine, ref in ((x1.grad, g[p + "grad_x1"]), (x2.grad, g[p + "grad_x2"])):
            np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(ref).max()))
    d = sphut.sphere_distance_torch(t(g["diag_x"]), t(g["diag_y"]), diag=True)
    np.testing.assert_allclose(d.cpu().numpy(), g["diag_dist"], rtol=1e-13)


def test_sphere_second_order_matches_plain_torch_autograd():
    """Exact Hessian-vector products (what pymanopt's TrustRegions asks of the sphere kernel) vs double backward through
    the plain formula in torch on the CPU."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((4, 5)); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.standard_normal((9, 5)); y /= np.linalg.norm(y, axis=1, keepdims=True)
    w = rng.standard_normal((4, 9)); u = rng.standard_normal((4, 5))
    for mode, beta in ((_lib.GABO_OUT_GAUSSIAN, 1.7), (_lib.GABO_OUT_LAPLACE, 0.9), (_lib.GABO_OUT_DISTANCE, 1.0)):
        def plain(xx):
            th = torch.acos((xx @ torch.tensor(y).T).clamp(-1 + 1e-15, 1 - 1e-15))
            if mode == _lib.GABO_OUT_DISTANCE:
                return th
            return torch.exp(-th * beta) if mode == _lib.GABO_OUT_LAPLACE else torch.exp(-th * th * beta)
        xc = torch.tensor(x, requires_grad=True)
        gc, = torch.autograd.grad((plain(xc) * torch.tensor(w)).sum(), xc, create_graph=True)
        hc, = torch.autograd.grad((gc * torch.tensor(u)).sum(), xc)
        xg = t(x, True)
        gg, = torch.autograd.grad((ops.sphere_kernel(xg, t(y), beta, mode) * t(w)).sum(), xg, create_graph=True)
        hg, = torch.autograd.grad((gg * t(u)).sum(), xg)
        np.testing.assert_allclose(gg.detach().cpu().numpy(), gc.detach().numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(hg.cpu().numpy(), hc.numpy(), rtol=1e-9, atol=1e-11)
    kl = SphereLaplaceKernel()
    kl.lengthscale = 1.3
    out = kl.forward(t(x), t(y))
    np.testing.assert_allclose(out.detach().cpu().numpy(), osph.sphere_laplace_kernel(x, y, 1 / kl.lengthscale.double().item() ** 2), rtol=1e-11)   # lengthscale is an fp32 parameter
    out.sum().backward()
    assert kl.raw_lengthscale.grad is not None and np.isfinite(kl.raw_lengthscale.grad.item())


# ------------------------------------------------------------------ Frobenius / log-Euclidean / nested kernels: input gradients
def test_log_euclidean_kernel_grads_golden(golden):
    from gabotorch_amd.kernel_utils.kernels_spd import SpdLogEuclideanGaussianKernel
    g = golden("nested_spd.npz")
    for d in (2, 3):
        x1, x2, ls = t(g[f"le{d}_x1"], True), t(g[f"le{d}_x2"], True), float(g[f"le{d}_ls"])
        kern = SpdLogEuclideanGaussianKernel().double()      # gpytorch's raw_lengthscale is fp32 by default
        kern.lengthscale = ls
        k = kern.forward(x1, x2)
        np.testing.assert_allclose(k.detach().cpu().numpy(), g[f"le{d}_K"], rtol=1e-10)
        (k * t(g[f"le{d}_gup"])).sum().backward()
        # reference = torch autograd through eig-based logm (spd_utils_torch.py:13-30) in fp64
        np.testing.assert_allclose(x1.grad.cpu().numpy(), g[f"le{d}_g1"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(x2.grad.cpu().numpy(), g[f"le{d}_g2"], rtol=1e-8, atol=1e-10)
        # lengthscale gradient against a central difference of the oracle kernel
        gl = kern.raw_lengthscale.grad
        assert gl is not None and torch.isfinite(gl).all()


@pytest.mark.parametrize("d", [2, 5, 10, 16, 32])
def test_log_euclidean_grads_vs_oracle(d):
    rng = np.random.default_rng(100 + d)
    n1, n2 = 37, 300            # n2 > 256: more than one column chunk in the backward kernel
    x1, x2 = rand_spd_mandel(rng, n1, d), rand_spd_mandel(rng, n2, d)
    x2[:3] = x1[:3]             # coincident points: zero distance, finite gradient
    ls = 1.7
    gup = rng.standard_normal((n1, n2))
    a, b = t(x1, True), t(x2, True)
    beta = torch.tensor(1.0 / ls ** 2, dtype=torch.float64, requires_grad=True)
    k = ops.frobenius_kernel(ops.spd_logm_mandel_diff(a), ops.spd_logm_mandel_diff(b), beta)
    np.testing.assert_allclose(k.detach().cpu().numpy(), ospd.log_euclidean_gaussian_kernel(x1, x2, ls), rtol=1e-9, atol=1e-14)
    (k * t(gup)).sum().backward()
    o1, o2 = ospd.log_euclidean_gaussian_kernel_grads(x1, x2, ls, gup)
    np.testing.assert_allclose(a.grad.cpu().numpy(), o1, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(o1).max()))
    np.testing.assert_allclose(b.grad.cpu().numpy(), o2, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(o2).max()))
    # d/dbeta by central difference on the oracle
    h = 1e-6
    kp = ospd.log_euclidean_gaussian_kernel(x1, x2, (1.0 / ls ** 2 + h) ** -0.5)
    km = ospd.log_euclidean_gaussian_kernel(x1, x2, (1.0 / ls ** 2 - h) ** -0.5)
    np.testing.assert_allclose(float(beta.grad), float((gup * (kp - km)).sum() / (2 * h)), rtol=1e-6)


def test_logm_adjoint_repeated_eigenvalues():
    # X = c I has all eigenvalues equal: autograd through eig() gives inf/nan there, the divided-difference form gives G / c
    d, c = 4, 2.5
    x = t(ospd.symmetric_matrix_to_vector_mandel(c * np.eye(d))[None], True)
    gy = np.random.default_rng(3).standard_normal((1, d * (d + 1) // 2))
    (ops.spd_logm_mandel_diff(x) * t(gy)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), gy / c, rtol=1e-12)


def test_frobenius_kernel_modes_by_finite_differences():
    rng = np.random.default_rng(8)
    d, n1, n2 = 3, 5, 7
    x1, x2 = rand_spd_mandel(rng, n1, d), rand_spd_mandel(rng, n2, d)
    gup = rng.standard_normal((n1, n2))
    for mode in (_lib.GABO_OUT_GAUSSIAN, _lib.GABO_OUT_LAPLACE, _lib.GABO_OUT_DISTANCE):
        a, b = t(x1, True), t(x2, True)
        (ops.frobenius_kernel(a, b, 0.3, mode) * t(gup)).sum().backward()
        f = lambda p, q: float((ops.frobenius_pairwise(t(p), t(q), 0.3, mode).cpu().numpy() * gup).sum())   # noqa: E731
        for arr, grad, which in ((x1, a.grad, 0), (x2, b.grad, 1)):
            num = np.zeros_like(arr)
            for idx in np.ndindex(arr.shape):
                hp, hm = arr.copy(), arr.copy()
                hp[idx] += 1e-6
                hm[idx] -= 1e-6
                num[idx] = (f(hp, x2) - f(hm, x2)) / 2e-6 if which == 0 else (f(x1, hp) - f(x1, hm)) / 2e-6
            np.testing.assert_allclose(grad.cpu().numpy(), num, rtol=2e-6, atol=1e-8)


def test_nested_kernels_differentiable_in_inputs_and_projection():
    from gabotorch_amd.kernel_utils.kernels_spd import NestedSpdLogEuclideanGaussianKernel, NestedSpdAffineInvariantGaussianKernel
    rng = np.random.default_rng(21)
    D, dl, n1, n2 = 6, 3, 9, 11
    x1, x2 = rand_spd_mandel(rng, n1, D), rand_spd_mandel(rng, n2, D)
    w = np.linalg.qr(rng.standard_normal((D, dl)))[0]
    gup = rng.standard_normal((n1, n2))

    def oracle_value(p, q, ww, ai):
        P = np.einsum("da,ndc,cb->nab", ww, ospd.vector_to_symmetric_matrix_mandel(p), ww)
        Q = np.einsum("da,ndc,cb->nab", ww, ospd.vector_to_symmetric_matrix_mandel(q), ww)
        pm, qm = ospd.symmetric_matrix_to_vector_mandel(P), ospd.symmetric_matrix_to_vector_mandel(Q)
        k = ospd.spd_ai_gaussian_kernel(pm, qm, 0.8) if ai else ospd.log_euclidean_gaussian_kernel(pm, qm, 1.3)
        return float((k * gup).sum())

    for ai in (False, True):
        if ai:
            kern = NestedSpdAffineInvariantGaussianKernel(D, dl, beta_min=0.1).double()
            kern.beta = 0.8
        else:
            kern = NestedSpdLogEuclideanGaussianKernel(D, dl).double()
            kern.lengthscale = 1.3
        kern.projection_matrix = torch.tensor(w)
        a, b = t(x1, True), t(x2, True)
        (kern.forward(a, b) * t(gup)).sum().backward()
        gw = kern.raw_projection_matrix.grad.cpu().numpy()
        h = 1e-6
        num_w = np.zeros_like(w)
        for idx in np.ndindex(w.shape):
            wp, wm = w.copy(), w.copy()
            wp[idx] += h
            wm[idx] -= h
            num_w[idx] = (oracle_value(x1, x2, wp, ai) - oracle_value(x1, x2, wm, ai)) / (2 * h)
        np.testing.assert_allclose(gw, num_w, rtol=1e-5, atol=1e-7)
        num_a = np.zeros_like(x1)
        for idx in np.ndindex(x1.shape):
            hp, hm = x1.copy(), x1.copy()
            hp[idx] += h
            hm[idx] -= h
            num_a[idx] = (oracle_value(hp, x2, w, ai) - oracle_value(hm, x2, w, ai)) / (2 * h)
        np.testing.assert_allclose(a.grad.cpu().numpy(), num_a, rtol=1e-5, atol=1e-7)
        assert torch.isfinite(b.grad).all()
