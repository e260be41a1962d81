"""Batched Riemannian operations (HIP) against the golden vectors of the reference's numpy maps and the CPU oracle."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, manifolds, ops
from oracle import spd as ospd
from oracle import sphere as osph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def rand_spd(rng, n, d, lo=0.2, hi=4.0):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def test_spd_maps_golden(golden):
    g = golden("spd_maps.npz")
    for d in (2, 3, 5):
        S, X, U = g[f"d{d}_S"], g[f"d{d}_X"], g[f"d{d}_log"]
        np.testing.assert_allclose(ops.spd_manifold_op(_lib.GABO_SPD_LOG, t(S), t(X)).cpu().numpy(), U, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(ops.spd_manifold_op(_lib.GABO_SPD_EXP, t(S), t(U)).cpu().numpy(), g[f"d{d}_explog"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(ops.spd_manifold_op(_lib.GABO_SPD_EXPM, t(g[f"d{d}_sym"])).cpu().numpy(), g[f"d{d}_multiexp"],
                                   rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(ops.spd_manifold_op(_lib.GABO_SPD_LOGM, t(X)).cpu().numpy(), g[f"d{d}_multilog"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(ops.spd_manifold_op(_lib.GABO_SPD_SQRTM, t(X)).cpu().numpy(), g[f"d{d}_sqrtm_torch"], rtol=1e-10, atol=1e-11)
        lam, grad = ops.spd_manifold_op(_lib.GABO_SPD_EIGMAX, t(X), want_grad=True)
        np.testing.assert_allclose(5.0 - lam.cpu().numpy(), g[f"d{d}_maxeig"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(-grad.cpu().numpy(), g[f"d{d}_maxeig_grad"], rtol=1e-9, atol=1e-10)
        lam = ops.spd_manifold_op(_lib.GABO_SPD_EIGMIN, t(X))
        np.testing.assert_allclose(lam.cpu().numpy() - 0.01, g[f"d{d}_mineig"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("d", [2, 5, 10, 20])
def test_spd_manifold_ops_vs_oracle(d):
    rng = np.random.default_rng(d)
    R = 37
    X, Y = rand_spd(rng, R, d), rand_spd(rng, R, d)
    U = rng.standard_normal((R, d, d)); U = 0.3 * (U + U.transpose(0, 2, 1))
    V = rng.standard_normal((R, d, d)); V = 0.3 * (V + V.transpose(0, 2, 1))
    G = rng.standard_normal((R, d, d))
    H = rng.standard_normal((R, d, d))
    man = manifolds.PositiveDefinite(d)
    np.testing.assert_allclose(man.exp(X, U), ospd.spd_exp(X, U), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(man.log(X, Y), ospd.spd_log(X, Y), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(man.exp(X, man.log(X, Y)), Y, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(man.inner(X, U, V), ospd.spd_inner(X, U, V), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(man.norm(X, U), ospd.spd_norm(X, U), rtol=1e-10)
    np.testing.assert_allclose(man.egrad2rgrad(X, G), ospd.spd_egrad2rgrad(X, G), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(man.ehess2rhess(X, G, H, U), ospd.spd_ehess2rhess(X, G, H, U), rtol=1e-11, atol=1e-11)
    dist = man.dist(X, Y)
    want = np.sqrt(np.maximum(np.diagonal(ospd.affine_invariant_distance(X, Y)) ** 2 - 1e-15, 0))
    np.testing.assert_allclose(dist, want, rtol=1e-9)
    # single point, torch tensors on the device
    one = man.exp(t(X[0]), t(U[0]))
    assert one.is_cuda and one.shape == (d, d)
    np.testing.assert_allclose(one.cpu().numpy(), ospd.spd_exp(X[0], U[0]), rtol=1e-9, atol=1e-10)
    with pytest.raises(RuntimeError, match="not positive definite"):
        man.exp(-np.eye(d), U[0])


def test_projection_and_log_euclid_golden(golden):
    g = golden("nested_spd.npz")
    y1 = ops.spd_project(t(g["x1_mandel"]), t(g["W"])).cpu().numpy()
    np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel(y1), g["Y1"], rtol=1e-11, atol=1e-12)
    y2 = ops.spd_project(t(g["x2_mandel"]), t(g["W"]))
    # nested affine-invariant kernel = projection then the pairwise kernel (kernels_nested_spd.py:122-136)
    dist = ops.spd_ai_pairwise(t(y1), y2, mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    np.testing.assert_allclose(dist, g["ai_dist"], rtol=5e-7, atol=5e-7)
    l1, l2 = ops.spd_logm_mandel(t(y1)), ops.spd_logm_mandel(y2)
    np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel(l1.cpu().numpy()), g["logY1"], rtol=1e-9, atol=1e-10)
    le = ops.frobenius_pairwise(l1, l2, mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    np.testing.assert_allclose(le, g["le_dist"], rtol=1e-9)
    x5 = ospd.symmetric_matrix_to_vector_mandel(g["X5"])
    np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel(ops.spd_project(t(x5), t(g["W5"])).cpu().numpy()), g["Y5"],
                               rtol=1e-11, atol=1e-12)
    # Frobenius kernel with the element-wise 1e-15 of the reference
    gm = golden("spd_maps.npz")
    for d in (2, 3, 5):
        a = ospd.symmetric_matrix_to_vector_mandel(gm[f"d{d}_S"])
        b = ospd.symmetric_matrix_to_vector_mandel(gm[f"d{d}_X"])
        np.testing.assert_allclose(ops.frobenius_pairwise(t(a), t(b), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy(), gm[f"d{d}_frob"], rtol=1e-12)
        k = ops.frobenius_pairwise(t(a), t(b), beta=0.37).cpu().numpy()
        np.testing.assert_allclose(k, np.exp(-0.37 * gm[f"d{d}_frob"] ** 2), rtol=1e-12)
    # a t-batch beyond the 65535 slices one launch carries (raw_samples x 1 x d candidates against a shared training set)
    rng = np.random.default_rng(8)
    cand, train = rng.standard_normal((70000, 1, 3)), rng.standard_normal((4, 3))
    got = ops.frobenius_pairwise(t(cand), t(train).expand(70000, 4, 3), beta=0.5).cpu().numpy()
    want = np.exp(-0.5 * ((cand - train[None] + 1e-15) ** 2).sum(-1))[:, None, :]
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_sphere_manifold_ops(golden):
    g = golden("sphere.npz")
    man = manifolds.Sphere(5)
    np.testing.assert_allclose(man.log(g["map_base"], g["map_x"]), g["map_log"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(man.exp(g["map_base"], g["map_log"]), g["map_exp"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(man.dist(g["map_base"], g["map_x"]), g["map_dist"], rtol=1e-11)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((50, 7)); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.standard_normal((50, 7)); y /= np.linalg.norm(y, axis=1, keepdims=True)
    h, eh, u = rng.standard_normal((3, 50, 7))
    man = manifolds.Sphere(7)
    np.testing.assert_allclose(man.proj(x, h), osph.proj(x, h), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(man.retr(x, 0.1 * h), osph.retr(x, 0.1 * h), rtol=1e-13)
    np.testing.assert_allclose(man.ehess2rhess(x, h, eh, u), osph.ehess2rhess(x, h, eh, u), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(man.transp(x, y, u), osph.transp(x, y, u), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(man.exp(x, np.zeros_like(x)), x, rtol=0, atol=0)
    np.testing.assert_allclose(man.log(x, x), 0.0, atol=1e-7)


def test_numpy_facing_utils_and_objectives(golden):
    """Reference-named numpy helpers (spd_utils / sphere_utils) and the benchmark objectives, against golden values produced by
    the reference's own functions."""
    from gabotorch_amd.BO_test_functions import test_functions as tf
    from gabotorch_amd.Riemannian_utils import sphere_utils, spd_utils
    g = golden("spd_maps.npz")
    S, X, U = g["d3_S"], g["d3_X"], g["d3_log"]
    np.testing.assert_allclose(spd_utils.logmap(X[0], S[0]), U[0], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(spd_utils.expmap(U[0], S[0]), g["d3_explog"][0], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(spd_utils.symmetric_matrix_to_vector_mandel(np.array([[2., .5], [.5, 1.]])), [2, 1, 0.5 * 2 ** 0.5])
    np.random.seed(int(g["sample_seed"]))
    man = manifolds.PositiveDefinite(5)
    man.min_eig, man.max_eig = 0.001, 5.0
    np.testing.assert_allclose(np.stack([spd_utils.spd_sample(man) for _ in range(3)]), g["sample_out"], rtol=1e-13, atol=1e-14)
    gs = golden("sphere.npz")
    np.testing.assert_allclose(sphere_utils.logmap(gs["map_x"][0], gs["map_base"][0])[:, 0], gs["map_log"][0], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(sphere_utils.expmap(gs["map_log"].T, gs["map_base"].T).T, gs["map_exp"], rtol=1e-11, atol=1e-13)
    go = golden("objectives.npz")
    for d in (2, 5):
        man = manifolds.PositiveDefinite(d)
        for k in range(5):
            x = torch.tensor(go[f"spd{d}_x"][k])
            np.testing.assert_allclose(tf.ackley_function_spd(x, man).item(), go[f"spd{d}_ackley"][k], rtol=1e-9)
            np.testing.assert_allclose(tf.rosenbrock_function_spd(x, man).item(), go[f"spd{d}_rosenbrock"][k], rtol=1e-8)
    for n in (3, 5):
        man = manifolds.Sphere(n)
        for k in range(5):
            np.testing.assert_allclose(tf.ackley_function_sphere(torch.tensor(go[f"sph{n}_x"][k]), man).item(),
                                       go[f"sph{n}_ackley"][k], rtol=1e-10)
    # known answers quoted in SURVEY App. A
    man2 = manifolds.PositiveDefinite(2)
    v = torch.tensor(spd_utils.symmetric_matrix_to_vector_mandel(np.array([[2., .5], [.5, 1.]])))
    np.testing.assert_allclose(tf.ackley_function_spd(v, man2).item(), 5.409645747183784, rtol=1e-10)
    np.testing.assert_allclose(tf.rosenbrock_function_spd(v, man2).item(), 533.8104003823937, rtol=1e-10)
    np.testing.assert_allclose(tf.ackley_function_sphere(torch.tensor([0., 1., 0.], dtype=torch.float64), manifolds.Sphere(3)).item(), 5.652422842950539, rtol=1e-10)


def test_nested_back_projection_golden(golden):
    from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_nested_spd_to_spd, projection_from_spd_to_nested_spd
    g = golden("nested_spd.npz")
    back = projection_from_nested_spd_to_spd(t(g["back_ylow"]), t(g["back_W"]), t(g["back_V"]), t(g["back_bottom"]), t(g["back_K"]))
    np.testing.assert_allclose(back.cpu().numpy(), g["back_X"], rtol=1e-10, atol=1e-11)
    # W^T X W recovers the latent matrix (it IS a right inverse of the projection)
    again = projection_from_spd_to_nested_spd(back, t(g["back_W"]))
    np.testing.assert_allclose(again.cpu().numpy(), g["back_ylow"], rtol=1e-9, atol=1e-10)
    one = projection_from_nested_spd_to_spd(t(g["back_ylow"][0]), t(g["back_W"]), t(g["back_V"]), t(g["back_bottom"]), t(g["back_K"]))
    assert one.shape == (5, 5)


def test_nested_sphere_projections_and_kernel_golden(golden):
    """f3: nested-sphere projection chain, back projection, NestedSphereGaussianKernel and its gradients with respect to the inputs
    AND the axes, against vectors produced by the reference (autograd)."""
    from gabotorch_amd.kernel_utils.kernels_nested_sphere import NestedSphereGaussianKernel
    from gabotorch_amd.nested_mappings import nested_spheres_utils as nsu
    from gabotorch_amd.Riemannian_utils.sphere_utils_torch import rotation_from_sphere_points_torch
    g = golden("nested_sphere.npz")
    for tag in "abc":
        nl = int(g[f"{tag}_nlevels"])
        dist = float(g[f"{tag}_dist"])
        x1 = torch.tensor(g[f"{tag}_x1"], device=DEV, requires_grad=True)
        x2 = torch.tensor(g[f"{tag}_x2"], device=DEV, requires_grad=True)
        axes = [torch.tensor(g[f"{tag}_axis{k}"], device=DEV, requires_grad=True) for k in range(nl)]
        dists = [torch.tensor([[dist]], dtype=torch.float64) for _ in range(nl)]
        dim = x1.shape[1]
        north = torch.zeros(1, dim, dtype=torch.float64, device=DEV)
        north[:, -1] = 1.0
        np.testing.assert_allclose(rotation_from_sphere_points_torch(axes[0].detach(), north).cpu().numpy(), g[f"{tag}_rot0"], atol=1e-14)
        np.testing.assert_allclose(nsu.projection_from_sphere_to_nested_sphere(x1.detach(), axes[0].detach(), dists[0]).cpu().numpy(),
                                   g[f"{tag}_nested0"], atol=1e-12)
        lv1 = nsu.projection_from_sphere_to_subsphere(x1, axes, dists)
        lv2 = nsu.projection_from_sphere_to_subsphere(x2, axes, dists)
        for k, lv in enumerate(lv1):
            np.testing.assert_allclose(lv.detach().cpu().numpy(), g[f"{tag}_level{k}"], atol=1e-12)
        back = nsu.projection_from_subsphere_to_sphere(lv1[-1].detach(), [a.detach() for a in axes], dists)
        for k, b in enumerate(back):
            np.testing.assert_allclose(b.cpu().numpy(), g[f"{tag}_back{k}"], atol=1e-12)
        beta = float(g[f"{tag}_beta"])
        K = ops.sphere_kernel(lv1[-1], lv2[-1], beta)
        np.testing.assert_allclose(K.detach().cpu().numpy(), g[f"{tag}_K"], rtol=1e-10)
        (K * torch.tensor(g[f"{tag}_gup"], device=DEV)).sum().backward()
        np.testing.assert_allclose(x1.grad.cpu().numpy(), g[f"{tag}_g1"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(x2.grad.cpu().numpy(), g[f"{tag}_g2"], rtol=1e-8, atol=1e-10)
        for k in range(nl):
            np.testing.assert_allclose(axes[k].grad.cpu().numpy(), g[f"{tag}_gaxis{k}"], rtol=1e-7, atol=1e-9)
    # the kernel class (distances fixed at pi/2 as in the reference): case "a" used pi/2
    kern = NestedSphereGaussianKernel(dim=5, latent_dim=3, beta_min=0.1).double()
    kern.axes = [torch.tensor(g[f"a_axis{k}"]) for k in range(2)]
    kern.beta = torch.tensor(float(g["a_beta"]), dtype=torch.float64)      # (a python float would pass through fp32, as in the reference)
    xa, xb = torch.tensor(g["a_x1"], device=DEV, requires_grad=True), torch.tensor(g["a_x2"], device=DEV)
    Kc = kern.forward(xa, xb)
    np.testing.assert_allclose(Kc.detach().cpu().numpy(), g["a_K"], rtol=1e-10)
    (Kc * torch.tensor(g["a_gup"], device=DEV)).sum().backward()
    np.testing.assert_allclose(xa.grad.cpu().numpy(), g["a_g1"], rtol=1e-8, atol=1e-10)
    assert kern.raw_axis_S5.grad is not None and kern.raw_beta.grad is not None
    np.testing.assert_allclose(kern.raw_axis_S5.grad.numpy(), g["a_gaxis0"] , rtol=1e-6, atol=1e-8)


def test_device_spd_sampler_distribution_and_reproducibility():
    """gabo_spd_sample: spd_sample's distribution (eigenvalues U[min, max], Haar eigenvectors) from a counter-based stream."""
    n, d, lo, hi = 20000, 5, 0.3, 4.0
    a = ops.spd_sample(n, d, lo, hi, seed=1234, device=DEV).cpu().numpy()
    np.testing.assert_array_equal(a, a.transpose(0, 2, 1))
    lam, vec = np.linalg.eigh(a)
    assert lam.min() >= lo - 1e-12 and lam.max() <= hi + 1e-12
    # uniform eigenvalues: mean (lo+hi)/2, variance (hi-lo)^2/12 ; pooled over all eigenvalues
    assert abs(lam.mean() - 0.5 * (lo + hi)) < 0.02 and abs(lam.var() - (hi - lo) ** 2 / 12) < 0.03
    # Haar eigenvectors: E[v_i^2] = 1/d for every coordinate, E[v_i v_j] = 0; take the eigenvector of the LARGEST eigenvalue
    v = vec[:, :, -1]
    np.testing.assert_allclose((v ** 2).mean(0), 1.0 / d, atol=0.01)
    np.testing.assert_allclose((v[:, 0] * v[:, 1]).mean(), 0.0, atol=0.01)
    # E[X] = mean eigenvalue * I
    np.testing.assert_allclose(a.mean(0), 0.5 * (lo + hi) * np.eye(d), atol=0.03)
    # same seed -> same matrices, whatever the batch size; another seed -> different
    b = ops.spd_sample(100, d, lo, hi, seed=1234, device=DEV).cpu().numpy()
    np.testing.assert_array_equal(b, a[:100])
    c = ops.spd_sample(100, d, lo, hi, seed=1235, device=DEV).cpu().numpy()
    assert np.abs(c - b).max() > 0.1
    # Mandel output is the same draw
    m = ops.spd_sample(100, d, lo, hi, seed=1234, device=DEV, mandel=True).cpu().numpy()
    np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel(m), b, atol=1e-14)
    # d = 1 and the largest supported d
    one = ops.spd_sample(1000, 1, lo, hi, seed=5, device=DEV).cpu().numpy()
    assert one.min() >= lo and one.max() <= hi
    big = ops.spd_sample(64, 16, lo, hi, seed=5, device=DEV).cpu().numpy()
    lam16 = np.linalg.eigvalsh(big)
    assert lam16.min() >= lo - 1e-10 and lam16.max() <= hi + 1e-10


@pytest.mark.parametrize("d", [5, 8, 9, 12, 13, 20, 21, 32])
def test_wave_eigen_solver_on_degenerate_and_scaled_inputs(d):
    """The wave-per-matrix Householder + QL solver (csrc/wave_eigh.hpp: every padded order and its padding cases) through the matrix
    functions and the extreme-eigenvalue op: identity, repeated eigenvalues, an exactly tridiagonal and an exactly diagonal matrix, the zero
    matrix (expm), a clustered spectrum, scales 1e-8 and 1e8, an indefinite input (expm of a tangent vector)."""
    rng = np.random.default_rng(100 + d)
    q = np.linalg.qr(rng.standard_normal((d, d)))[0]
    spectra = [np.ones(d), np.repeat([0.5, 2.0], [d // 2, d - d // 2]), np.linspace(0.3, 3.0, d),
               1.0 + 1e-9 * np.arange(d), np.concatenate([[1e-3], np.full(d - 1, 4.0)])]
    mats = [(q * lam) @ q.T for lam in spectra]
    mats.append(np.diag(np.linspace(0.2, 2.0, d)))
    tri = np.diag(np.linspace(1.0, 2.0, d)) + np.diag(np.full(d - 1, 0.3), 1) + np.diag(np.full(d - 1, 0.3), -1)
    mats += [tri, 1e-8 * mats[2], 1e8 * mats[2]]
    mats = np.stack([0.5 * (m + m.T) for m in mats])
    lam, vec = np.linalg.eigh(mats)
    fun = lambda f: np.einsum("nab,nb,ncb->nac", vec, f(lam), vec)      # noqa: E731
    def worst(got, want, scale):            # per matrix: max abs error in units of `scale`
        return np.abs(got - want).max(axis=(1, 2)) / scale
    err = worst(ops.spd_manifold_op(_lib.GABO_SPD_LOGM, t(mats)).cpu().numpy(), fun(np.log), np.abs(np.log(lam)).max(axis=1) + 0.1)
    assert err.max() < 1e-12, err
    err = worst(ops.spd_manifold_op(_lib.GABO_SPD_SQRTM, t(mats)).cpu().numpy(), fun(np.sqrt), np.sqrt(lam.max(axis=1)))
    assert err.max() < 1e-13, err
    for op, pick in ((_lib.GABO_SPD_EIGMAX, -1), (_lib.GABO_SPD_EIGMIN, 0)):
        val, vv = ops.spd_manifold_op(op, t(mats), want_grad=True)
        assert (np.abs(val.cpu().numpy() - lam[:, pick]) <= 1e-14 * np.abs(lam).max(axis=1)).all()      # (absolute accuracy eps |A|: the small end of a wide spectrum keeps fewer digits)
        vv = vv.cpu().numpy()                                           # v v^T of SOME unit vector of the (possibly degenerate) eigenspace
        np.testing.assert_allclose(np.trace(vv, axis1=1, axis2=2), 1.0, rtol=1e-12)
        resid = np.einsum("nab,nbc->nac", mats, vv) - lam[:, pick, None, None] * vv
        np.testing.assert_array_less(np.abs(resid).max(axis=(1, 2)), 1e-11 * np.abs(lam).max(axis=1) + 1e-300)
    tangent = np.stack([np.zeros((d, d)), 0.5 * (mats[2] - 1.5 * np.eye(d))])       # zero and an indefinite symmetric matrix
    lt, vt = np.linalg.eigh(tangent)
    err = worst(ops.spd_manifold_op(_lib.GABO_SPD_EXPM, t(tangent)).cpu().numpy(), np.einsum("nab,nb,ncb->nac", vt, np.exp(lt), vt), np.exp(lt.max(axis=1)))
    assert err.max() < 1e-13, err


@pytest.mark.parametrize("d", [5, 7, 8, 9, 12, 14, 16, 17, 20, 23, 24, 28, 32])
def test_lane_group_eigen_solver_on_many_spectra(d):
    """Round 4: for orders 7 ... 32 the eigenpairs come from the lane-group solver of csrc/wave_eigh.hpp (multisection on Sturm counts,
    Rayleigh-quotient iteration per lane, windowed Newton-Schulz step), with the QL path as its fallback.  1200 matrices per order: Wishart
    spectra (crowded small end), spectra over 8 decades, pairs of eigenvalues 1e-3 ... 1e-12 apart (in and below the solver's isolation
    threshold: both routes), a cluster of five, graded tridiagonal-like matrices; matrix square root and logarithm against numpy's eigh,
    and sqrtm(A)^2 = A."""
    rng = np.random.default_rng(500 + d)
    n = 200
    mats = []
    g = rng.standard_normal((n, d, d))
    mats.append(np.einsum("nab,ncb->nac", g, g) / d + 0.1 * np.eye(d))
    def with_spectrum(lam):
        q = np.linalg.qr(rng.standard_normal((lam.shape[0], d, d)))[0]
        return np.einsum("nab,nb,ncb->nac", q, lam, q)
    mats.append(with_spectrum(10.0 ** rng.uniform(-4, 4, (n, d))))
    lam = rng.uniform(0.5, 3.0, (n, d))
    lam[:, 1] = lam[:, 0] * (1.0 + 10.0 ** rng.uniform(-12, -3, n))
    mats.append(with_spectrum(lam))
    lam = rng.uniform(0.5, 3.0, (n, d))
    lam[:, :5] = lam[:, :1] * (1.0 + 10.0 ** rng.uniform(-9, -4, (n, 1)) * np.arange(5))
    mats.append(with_spectrum(lam))
    grade = 10.0 ** (-np.arange(d) * rng.uniform(0.05, 0.25, (n, 1)))          # (entries down to 1e-11 of the largest)
    mats.append(np.einsum("na,nab,nb->nab", grade, mats[0], grade) + 1e-9 * np.eye(d))
    mats.append(with_spectrum(rng.uniform(0.2, 5.0, (n, d))) * 10.0 ** rng.uniform(-6, 6, (n, 1, 1)))
    mats = np.concatenate(mats)
    mats = 0.5 * (mats + mats.transpose(0, 2, 1))
    lam, vec = np.linalg.eigh(mats)
    assert lam.min() > 0
    fun = lambda f: np.einsum("nab,nb,ncb->nac", vec, f(lam), vec)      # noqa: E731
    root = ops.spd_manifold_op(_lib.GABO_SPD_SQRTM, t(mats)).cpu().numpy()
    scale = np.sqrt(lam.max(axis=1))
    err = np.abs(root - fun(np.sqrt)).max(axis=(1, 2)) / scale
    # (the small eigenvalues are known to eps |A|, their square roots to eps sqrt(cond) / 2 of sqrt |A| - in either solver)
    cond = lam.max(axis=1) / lam.min(axis=1)
    assert (err < 2e-13 + 1e-15 * np.sqrt(cond)).all(), (err / (2e-13 + 1e-15 * np.sqrt(cond))).max()
    back = np.abs(np.einsum("nab,nbc->nac", root, root) - mats).max(axis=(1, 2)) / lam.max(axis=1)
    assert back.max() < 2e-13, (back.max(), back.argmax())
    logm = ops.spd_manifold_op(_lib.GABO_SPD_LOGM, t(mats)).cpu().numpy()
    err = np.abs(logm - fun(np.log)).max(axis=(1, 2)) / (np.abs(np.log(lam)).max(axis=1) + 0.1)
    # (log of a spectrum over many decades: the small eigenvalues are known to eps |A| only, their logarithms to eps cond(A))
    assert (err < 1e-13 + 4e-16 * cond).all(), (err / (1e-13 + 4e-16 * cond)).max()
