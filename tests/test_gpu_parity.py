"""Parity of the HIP path (through the C ABI) against the golden vectors and the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, ops
from oracle import spd as ospd
from oracle import sphere as osph

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
F32 = 5e-7   # fp32 eigenvalue sink of the reference (spd_utils_torch.py:108): golden SPD distances are only this good
RTOL = 1e-5  # BASELINE.json north_star tolerance (relative, fp64)


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def rand_spd_mandel(rng, n, d, lo=0.05, hi=5.0):
    out = np.empty((n, d, d))
    for k in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        m = (q * rng.uniform(lo, hi, d)) @ q.T
        out[k] = 0.5 * (m + m.T)
    return ospd.symmetric_matrix_to_vector_mandel(out)


def test_native_library_is_loaded():
    lib = _lib.load()
    assert lib.gabo_version() >= 100
    assert torch.cuda.is_available()


def test_mandel_golden(golden):
    g = golden("mandel.npz")
    for d in (2, 3, 5, 10, 20):
        m = ops.mandel_to_matrix(t(g[f"d{d}_vec"])).cpu().numpy()
        np.testing.assert_array_equal(m, g[f"d{d}_mat"])                      # bit exact: same division by 2**0.5
        v = ops.matrix_to_mandel(t(g[f"d{d}_nonsym"])).cpu().numpy()
        np.testing.assert_allclose(v, g[f"d{d}_nonsym_vec"], rtol=1e-15, atol=1e-16)
        np.testing.assert_allclose(ops.matrix_to_mandel(t(m)).cpu().numpy(), g[f"d{d}_vec"], rtol=1e-15, atol=1e-16)


def test_spd_ai_golden(golden):
    g = golden("spd_ai.npz")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        x1, x2, beta = g[p + "x1"], g[p + "x2"], float(g[p + "beta"])
        dist = ops.spd_ai_pairwise(t(x1), t(x2), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
        assert dist.shape == g[p + "dist"].shape
        np.testing.assert_allclose(dist, g[p + "dist"], rtol=F32, atol=F32)
        if p + "dist_np" in g:    # the reference's independent fp64 statement: tight
            np.testing.assert_allclose(dist, np.sqrt(g[p + "dist_np"] ** 2 + 1e-15), rtol=1e-10, atol=1e-11)
        k = ops.spd_ai_pairwise(t(x1), t(x2), beta=beta).cpu().numpy()
        np.testing.assert_allclose(k, g[p + "K"], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(k, ospd.spd_ai_gaussian_kernel(x1, x2, beta), rtol=1e-9, atol=1e-12)
    d = ops.spd_ai_pairwise(t(ospd.symmetric_matrix_to_vector_mandel(g["kat_a"])),
                            t(ospd.symmetric_matrix_to_vector_mandel(g["kat_b"])), mode=_lib.GABO_OUT_DISTANCE)
    np.testing.assert_allclose(d.item(), 1.4033966394735078, rtol=1e-12)
    i10 = ospd.symmetric_matrix_to_vector_mandel(np.eye(10)[None])
    d = ops.spd_ai_pairwise(t(i10), t(np.e * i10), mode=_lib.GABO_OUT_DISTANCE)
    np.testing.assert_allclose(d.item(), 10 ** 0.5, rtol=1e-13)
    d = ops.spd_ai_pairwise(t(i10), t(i10), mode=_lib.GABO_OUT_DISTANCE)
    np.testing.assert_allclose(d.item(), 1e-15 ** 0.5, rtol=1e-6)            # d(X,X) = sqrt(1e-15), not 0


def test_letters_fixture(golden):
    g = golden("letters_spd2.npz")
    x = t(g["x_mandel"])
    k = ops.spd_ai_pairwise(x, x, beta=float(g["beta"])).cpu().numpy()
    np.testing.assert_allclose(k, g["K"], rtol=RTOL, atol=1e-7)
    ks = ops.spd_ai_pairwise(x, x, beta=float(g["beta"]), symmetric=True).cpu().numpy()
    np.testing.assert_allclose(ks, k, rtol=1e-10, atol=1e-13)
    np.testing.assert_array_equal(ks, ks.T)


@pytest.mark.parametrize("d", list(range(2, 33)))          # (every dimension the library takes: each is its own kernel instantiation up to 20)
def test_spd_ai_vs_oracle_all_dims(d):
    rng = np.random.default_rng(d)
    x1 = rand_spd_mandel(rng, 37, d)
    x2 = rand_spd_mandel(rng, 70, d)
    want = ospd.affine_invariant_distance(ospd.vector_to_symmetric_matrix_mandel(x1),
                                          ospd.vector_to_symmetric_matrix_mandel(x2))
    got = ops.spd_ai_pairwise(t(x1), t(x2), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)
    lap = ops.spd_ai_pairwise(t(x1), t(x2), beta=0.7, mode=_lib.GABO_OUT_LAPLACE).cpu().numpy()
    np.testing.assert_allclose(lap, np.exp(-0.7 * want), rtol=1e-10)


def test_spd_ai_gaussian_d2_kernel():
    """d = 2 Gaussian values without a distance output run in their own kernel (spd_ai_gauss2_kernel): ragged sizes, batches, a shared
    set, the x1-is-x2 build, near-identical and badly conditioned pairs, and agreement with the general kernel's distance output."""
    rng = np.random.default_rng(22)
    for n1, n2 in [(1, 1), (3, 64), (17, 257), (70, 300), (130, 65)]:
        x1 = np.stack([rand_spd_mandel(rng, n1, 2) for _ in range(2)])
        x2 = np.stack([rand_spd_mandel(rng, n2, 2) for _ in range(2)])
        got = ops.spd_ai_pairwise(t(x1), t(x2), beta=0.8).cpu().numpy()
        np.testing.assert_allclose(got, ospd.spd_ai_gaussian_kernel(x1, x2, 0.8), rtol=1e-9, atol=1e-13)
        dist = ops.spd_ai_pairwise(t(x1), t(x2), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
        np.testing.assert_allclose(got, np.exp(-0.8 * dist ** 2), rtol=1e-11, atol=1e-300)
    x = rand_spd_mandel(rng, 333, 2)
    X = t(x)
    sym = ops.spd_ai_pairwise(X, X, beta=1.1, symmetric=True).cpu().numpy()
    full = ops.spd_ai_pairwise(X, X.clone(), beta=1.1).cpu().numpy()
    np.testing.assert_array_equal(sym, sym.T)
    np.testing.assert_allclose(sym, full, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(sym, ospd.spd_ai_gaussian_kernel(x, x, 1.1), rtol=1e-9, atol=1e-13)
    shared = t(x[:50]).expand(4, 50, 3)
    cand = np.stack([rand_spd_mandel(rng, 1, 2) for _ in range(4)])
    got = ops.spd_ai_pairwise(t(cand), shared, beta=0.5).cpu().numpy()
    np.testing.assert_allclose(got, ospd.spd_ai_gaussian_kernel(cand, np.broadcast_to(x[:50], (4, 50, 3)), 0.5), rtol=1e-9, atol=1e-13)
    # near-identical pairs (K -> 1) and condition numbers up to e^18
    a = rand_spd_mandel(rng, 64, 2)
    b = a * (1.0 + 1e-9 * np.arange(64))[:, None]
    got = ops.spd_ai_pairwise(t(a), t(b), beta=2.0).cpu().numpy()
    np.testing.assert_allclose(got, ospd.spd_ai_gaussian_kernel(a, b, 2.0), rtol=1e-12, atol=0)
    q = np.linalg.qr(rng.standard_normal((32, 2, 2)))[0]
    ill = np.einsum("nab,nb,ncb->nac", q, np.exp(rng.uniform(-9, 9, (32, 2))), q)
    illv = ospd.symmetric_matrix_to_vector_mandel(0.5 * (ill + ill.transpose(0, 2, 1)))
    got = ops.spd_ai_pairwise(t(a), t(illv), beta=0.01).cpu().numpy()
    np.testing.assert_allclose(got, ospd.spd_ai_gaussian_kernel(a, illv, 0.01), rtol=1e-6, atol=1e-300)
    # tau = cosh(delta) beyond the kernel's acosh^2 table (eigenvalue ratio of M beyond e^18: tau >= 2^12) - the lanes that leave the table take
    # acosh from log + sqrt behind a wave-uniform branch; mixed with ordinary pairs in the same wave.  Well-conditioned by construction
    # (diagonal matrices: the Cholesky factors are exact), so the comparison is tight
    lam = np.exp(np.concatenate([rng.uniform(-14, 14, (40, 2)), rng.uniform(-1, 1, (40, 2))]))
    far = np.zeros((80, 3))
    far[:, 0], far[:, 1] = lam[:, 0], lam[:, 1]
    eye = np.tile(np.array([[1.0, 1.0, 0.0]]), (5, 1))
    want = np.exp(-0.003 * (np.log(lam[:, 0]) ** 2 + np.log(lam[:, 1]) ** 2 + 1e-15))
    assert float(np.max(np.abs(np.log(lam[:40, 0] / lam[:40, 1])))) > 20          # delta > 10: far outside the table
    got = ops.spd_ai_pairwise(t(eye), t(far), beta=0.003).cpu().numpy()
    np.testing.assert_allclose(got, np.broadcast_to(want, (5, 80)), rtol=1e-12, atol=0)
    np.testing.assert_allclose(got, ospd.spd_ai_gaussian_kernel(eye, far, 0.003), rtol=1e-9, atol=0)


def test_spd_ai_shapes_batches_and_edges():
    rng = np.random.default_rng(5)
    d = 5
    # batched, ragged sizes around the 64/256 tile edges, and the expand()ed train set gpytorch hands over
    for n1, n2 in [(1, 1), (1, 63), (3, 64), (2, 65), (17, 257), (5, 300)]:
        x1 = np.stack([rand_spd_mandel(rng, n1, d) for _ in range(3)])
        x2 = np.stack([rand_spd_mandel(rng, n2, d) for _ in range(3)])
        want = ospd.spd_ai_gaussian_kernel(x1, x2, 0.4)
        got = ops.spd_ai_pairwise(t(x1), t(x2), beta=0.4)
        assert got.shape == (3, n1, n2)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-9, atol=1e-13)
    train = rand_spd_mandel(rng, 11, d)
    cand = np.stack([rand_spd_mandel(rng, 1, d) for _ in range(40)])           # b x 1 x d_vec
    x2 = t(train).expand(40, 11, train.shape[-1])                             # stride-0 batch
    got = ops.spd_ai_pairwise(t(cand), x2, beta=0.9).cpu().numpy()
    want = ospd.spd_ai_gaussian_kernel(cand, np.broadcast_to(train, (40,) + train.shape), 0.9)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-13)
    # two batch dims
    x1 = np.stack([rand_spd_mandel(rng, 4, 3) for _ in range(6)]).reshape(2, 3, 4, 6)
    x2 = np.stack([rand_spd_mandel(rng, 5, 3) for _ in range(6)]).reshape(2, 3, 5, 6)
    got = ops.spd_ai_pairwise(t(x1), t(x2), beta=0.3).cpu().numpy()
    np.testing.assert_allclose(got, ospd.spd_ai_gaussian_kernel(x1, x2, 0.3), rtol=1e-9, atol=1e-13)
    # empty
    assert ops.spd_ai_pairwise(t(np.zeros((0, 15))), t(rand_spd_mandel(rng, 3, 5))).shape == (0, 3)
    # CPU tensors in -> CPU tensor out (the reference's examples build CPU tensors)
    k = ops.spd_ai_pairwise(torch.tensor(train), torch.tensor(train), beta=0.5)
    assert k.device.type == "cpu" and k.dtype == torch.float64


def test_spd_not_spd_is_reported(raising):
    bad = np.array([[1.0, 1.0, 3.0 * 2 ** 0.5]])       # [[1,3],[3,1]]: indefinite
    good = ospd.symmetric_matrix_to_vector_mandel(np.eye(2)[None])
    with raising("not positive definite"):
        ops.spd_ai_pairwise(t(bad), t(good))
    # an indefinite matrix in the SECOND set: the reference factors x1 only - the pair goes through symeig and log and comes out NaN, without an
    # exception (spd_utils_torch.py:87, 109-120; tests/test_gpu_nan.py)
    assert bool(torch.isnan(ops.spd_ai_pairwise(t(good), t(bad))).all())
    with raising("not positive definite"):
        ops.spd_ai_pairwise(t(good), t(np.array([[1.0, float("nan"), 0.0]])))
    with pytest.raises(RuntimeError, match="unsupported dimension"):
        ops.spd_ai_pairwise(t(np.ones((2, 33 * 34 // 2))), t(np.ones((2, 33 * 34 // 2))))


def test_spd_properties_at_scale():
    """Size-independent properties on a larger set: symmetry, unit diagonal, affine invariance, inversion invariance."""
    rng = np.random.default_rng(77)
    d, n = 10, 512
    x = rand_spd_mandel(rng, n, d)
    X = t(x)
    D = ops.spd_ai_pairwise(X, X, mode=_lib.GABO_OUT_DISTANCE)
    np.testing.assert_allclose(D.cpu().numpy(), D.T.cpu().numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(torch.diagonal(D).cpu().numpy(), 1e-15 ** 0.5, rtol=1e-3)
    Ds = ops.spd_ai_pairwise(X, X, mode=_lib.GABO_OUT_DISTANCE, symmetric=True)
    np.testing.assert_array_equal(Ds.cpu().numpy(), Ds.T.cpu().numpy())
    np.testing.assert_allclose(Ds.cpu().numpy(), D.cpu().numpy(), rtol=1e-9, atol=1e-10)
    # d(A X A^T, A Y A^T) = d(X, Y)
    A = rng.standard_normal((d, d)) + 3 * np.eye(d)
    M = ospd.vector_to_symmetric_matrix_mandel(x[:64])
    Mc = A @ M @ A.T
    xc = ospd.symmetric_matrix_to_vector_mandel(0.5 * (Mc + Mc.transpose(0, 2, 1)))
    Dc = ops.spd_ai_pairwise(t(xc), t(xc), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    np.testing.assert_allclose(Dc, D[:64, :64].cpu().numpy(), rtol=1e-7, atol=1e-8)
    # d(X^-1, Y^-1) = d(X, Y)
    Mi = np.linalg.inv(M)
    xi = ospd.symmetric_matrix_to_vector_mandel(0.5 * (Mi + Mi.transpose(0, 2, 1)))
    Di = ops.spd_ai_pairwise(t(xi), t(xi), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    np.testing.assert_allclose(Di, D[:64, :64].cpu().numpy(), rtol=1e-7, atol=1e-8)
    # oracle on a sub-block
    want = ospd.affine_invariant_distance(ospd.vector_to_symmetric_matrix_mandel(x[:48]),
                                          ospd.vector_to_symmetric_matrix_mandel(x[100:164]))
    np.testing.assert_allclose(D[:48, 100:164].cpu().numpy(), want, rtol=1e-10, atol=1e-12)


def test_sphere_golden(golden):
    g = golden("sphere.npz")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        x1, x2, beta = g[p + "x1"], g[p + "x2"], float(g[p + "beta"])
        dist = ops.sphere_pairwise(t(x1), t(x2), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
        # acos near +-1 amplifies 1-ulp differences of the inner product to ~1e-8: absolute tolerance there
        np.testing.assert_allclose(dist, g[p + "dist"], rtol=1e-12, atol=3e-8)
        k = ops.sphere_pairwise(t(x1), t(x2), beta=beta).cpu().numpy()
        np.testing.assert_allclose(k, g[p + "K"], rtol=1e-11, atol=1e-14)
    got = ops.sphere_pairwise(t(g["diag_x"]), t(g["diag_y"]), mode=_lib.GABO_OUT_DISTANCE, diag=True).cpu().numpy()
    np.testing.assert_allclose(got, g["diag_dist"], rtol=1e-13)
    e = np.eye(3)
    assert ops.sphere_pairwise(t(e[0:1]), t(e[1:2]), mode=_lib.GABO_OUT_DISTANCE).item() == np.pi / 2
    np.testing.assert_allclose(ops.sphere_pairwise(t(e[0:1]), t(-e[0:1]), mode=_lib.GABO_OUT_DISTANCE).item(),
                               g["kat_e1me1"].item(), rtol=1e-15)
    np.testing.assert_allclose(ops.sphere_pairwise(t(e[0:1]), t(e[0:1]), mode=_lib.GABO_OUT_DISTANCE).item(),
                               4.4703483581542975e-08, rtol=1e-9)


@pytest.mark.parametrize("dim,n1,n2,batch", [(3, 50, 70, ()), (10, 300, 513, ()), (51, 9, 130, (2,)), (101, 7, 5, (2, 2))])
def test_sphere_vs_oracle(dim, n1, n2, batch):
    rng = np.random.default_rng(dim)
    x1 = rng.standard_normal(batch + (n1, dim))
    x1 /= np.linalg.norm(x1, axis=-1, keepdims=True)
    x2 = rng.standard_normal(batch + (n2, dim))
    x2 /= np.linalg.norm(x2, axis=-1, keepdims=True)
    for mode, beta, want in [(_lib.GABO_OUT_GAUSSIAN, 1.3, osph.sphere_gaussian_kernel(x1, x2, 1.3)),
                             (_lib.GABO_OUT_LAPLACE, 0.8, osph.sphere_laplace_kernel(x1, x2, 0.8)),
                             (_lib.GABO_OUT_DISTANCE, 1.0, osph.sphere_distance(x1, x2))]:
        got = ops.sphere_pairwise(t(x1), t(x2), beta=beta, mode=mode).cpu().numpy()
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("dim", [1, 2, 4, 5, 8, 12, 13, 16, 17])
def test_sphere_every_k_step_count(dim):
    """dim <= 16 runs with the x2 fragments in registers and the x1 rows in LDS (1..4 MFMA K steps, with and without K padding), dim > 16
    with per-chunk operand loads; ragged sizes exercise the predicated edge tiles, beta >= 1000 the unscaled coefficients."""
    rng = np.random.default_rng(100 + dim)
    for n1, n2 in ((77, 333), (130, 64), (1, 1), (64, 257)):
        x1 = rng.standard_normal((n1, dim)); x1 /= np.linalg.norm(x1, axis=-1, keepdims=True)
        x2 = rng.standard_normal((n2, dim)); x2 /= np.linalg.norm(x2, axis=-1, keepdims=True)
        if dim > 1 and n1 > 1:
            x2[0] = x1[0]                       # an identical and an antipodal pair: the clamp at both ends
            x2[-1] = -x1[1]
        for beta in (0.37, 1500.0):
            got = ops.sphere_pairwise(t(x1), t(x2), beta=beta).cpu().numpy()
            np.testing.assert_allclose(got, osph.sphere_gaussian_kernel(x1, x2, beta), rtol=1e-11 * max(1.0, beta), atol=1e-300)
        got = ops.sphere_pairwise(t(x1), t(x2), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
        np.testing.assert_allclose(got, osph.sphere_distance(x1, x2), rtol=1e-12, atol=3e-8)


def test_sphere_batched_register_path():
    """leading batch dimensions (and a set shared by the whole batch: stride 0) through the dim <= 16 path"""
    rng = np.random.default_rng(77)
    x1 = rng.standard_normal((3, 2, 45, 10)); x1 /= np.linalg.norm(x1, axis=-1, keepdims=True)
    x2 = rng.standard_normal((3, 2, 70, 10)); x2 /= np.linalg.norm(x2, axis=-1, keepdims=True)
    got = ops.sphere_pairwise(t(x1), t(x2), beta=0.9).cpu().numpy()
    np.testing.assert_allclose(got, osph.sphere_gaussian_kernel(x1, x2, 0.9), rtol=1e-11, atol=1e-14)
    shared = t(x2[0, 0]).expand(3, 2, 70, 10)
    got = ops.sphere_pairwise(t(x1), shared, beta=0.9).cpu().numpy()
    np.testing.assert_allclose(got, osph.sphere_gaussian_kernel(x1, np.broadcast_to(x2[0, 0], x2.shape), 0.9), rtol=1e-11, atol=1e-14)


def test_sphere_symmetric_mode():
    rng = np.random.default_rng(8)
    for n in (1, 17, 256, 700):
        x = rng.standard_normal((n, 6)); x /= np.linalg.norm(x, axis=1, keepdims=True)
        X = t(x)
        full = ops.sphere_pairwise(X, X, beta=2.1).cpu().numpy()
        sym = ops.sphere_pairwise(X, X, beta=2.1, symmetric=True).cpu().numpy()
        np.testing.assert_array_equal(sym, sym.T)
        np.testing.assert_allclose(sym, full, rtol=1e-15, atol=1e-16)
        np.testing.assert_allclose(sym, osph.sphere_gaussian_kernel(x, x, 2.1), rtol=1e-11, atol=1e-14)


def test_baseline_config3_full_size_properties():
    """BASELINE.json config 3 at full size (N=4096, d=10): size-independent properties + an oracle sub-block."""
    import bench
    x = bench.synthetic_spd_mandel(4096, 10, 1234)
    X = t(x)
    K = ops.spd_ai_pairwise(X, X, beta=bench.BETA)
    Ks = ops.spd_ai_pairwise(X, X, beta=bench.BETA, symmetric=True)
    assert K.shape == (4096, 4096) and bool(torch.isfinite(K).all())
    assert float((K - K.T).abs().max()) < 1e-10                    # d(A,B) = d(B,A) through two different Cholesky factors
    assert torch.equal(Ks, Ks.T)
    assert float((K - Ks).abs().max()) < 1e-10
    assert float((torch.diagonal(K) - 1).abs().max()) < 1e-12      # exp(-beta 1e-15)
    assert float(K.max()) <= 1.0 and float(K.min()) >= 0.0
    # transposed evaluation order: K(X[a], X[b]) == K(X[b], X[a])^T on ragged slices that do not align with the tiles
    a, b = slice(37, 1000), slice(2111, 4096)
    np.testing.assert_allclose(ops.spd_ai_pairwise(X[a], X[b], beta=bench.BETA).cpu().numpy(),
                               ops.spd_ai_pairwise(X[b], X[a], beta=bench.BETA).T.cpu().numpy(), rtol=1e-9, atol=1e-13)
    # a slice evaluated on its own groups other pairs into a wave: the look-ahead QL (spd_eig.hpp) lets a lane spend the sweeps its wave
    # still needs on its own next stage, so the result depends on the wave's company in the last bits (equally accurate, not bitwise)
    np.testing.assert_allclose(ops.spd_ai_pairwise(X[a], X[b], beta=bench.BETA).cpu().numpy(), K[a, b].cpu().numpy(), rtol=1e-11, atol=1e-300)
    # ... and the same call is deterministic
    assert torch.equal(ops.spd_ai_pairwise(X[a], X[b], beta=bench.BETA), ops.spd_ai_pairwise(X[a], X[b], beta=bench.BETA))
    blk = (slice(4000, 4096), slice(0, 130))
    want = ospd.spd_ai_gaussian_kernel(x[blk[0]], x[blk[1]], bench.BETA)
    np.testing.assert_allclose(K[blk].cpu().numpy(), want, rtol=1e-9, atol=1e-14)


def test_baseline_config2_full_size_properties():
    """BASELINE.json config 2 (S^9, N=4096)."""
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((4096, 10)); x /= np.linalg.norm(x, axis=1, keepdims=True)
    X = t(x)
    K = ops.sphere_pairwise(X, X, beta=0.6 + np.log(2))
    assert torch.equal(K, K.T)                                      # <x,y> is accumulated in the same order both ways
    np.testing.assert_allclose(torch.diagonal(K).cpu().numpy(), np.exp(-(0.6 + np.log(2)) * np.arccos(1 - 1e-15) ** 2), rtol=1e-9)
    np.testing.assert_allclose(K[100:164, 4000:4096].cpu().numpy(), osph.sphere_gaussian_kernel(x[100:164], x[4000:4096], 0.6 + np.log(2)),
                               rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("dim", [3, 7, 10, 16])
@pytest.mark.parametrize("beta", [0.2, 1.2931471805599454, 4.0, 4.5])
def test_sphere_large_gram_kernel_value_table(dim, beta):
    """Gram matrices of >= 4 M outputs with beta <= 4 take the kernel-value table (sphere_gauss_finish_kt: blocks of 1024 threads, no exp per
    output); beta = 4.5 the round-3 epilogue on the same inputs.  Ragged sizes (partial column block, partial row chunk), the clamp edge
    cases (identical / antipodal / nearly identical points, kernels_sphere.py:90-94 with sphere_utils_torch.py:53) and the x1-is-x2 build."""
    rng = np.random.default_rng(dim)
    n1, n2 = 2050, 2100
    x1 = rng.standard_normal((n1, dim)); x1 /= np.linalg.norm(x1, axis=1, keepdims=True)
    x2 = rng.standard_normal((n2, dim)); x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
    x2[:40] = x1[:40]                                   # identical: d = 4.47e-8
    x2[40:80] = -x1[40:80]                              # antipodal
    y = x1[80:120] + 1e-9 * rng.standard_normal((40, dim))
    x2[80:120] = y / np.linalg.norm(y, axis=1, keepdims=True)
    want = osph.sphere_gaussian_kernel(x1, x2, beta)
    got = ops.sphere_pairwise(t(x1), t(x2), beta=beta).cpu().numpy()
    # (relative to K: at theta -> pi the conditioning of exp(-beta theta^2) in c is 2 beta pi / sqrt(1 - c^2) - numpy's own arccos / exp chain
    # is only this good there, tools/sim/gen_sphere_ktab.py)
    np.testing.assert_allclose(got, want, rtol=5e-12, atol=1e-300)
    ws = osph.sphere_gaussian_kernel(x2, x2, beta)
    gs = ops.sphere_pairwise(t(x2), t(x2), beta=beta, symmetric=True).cpu().numpy()
    np.testing.assert_allclose(gs, ws, rtol=5e-12, atol=1e-300)
    np.testing.assert_array_equal(gs, gs.T)
    # a NaN row / column in the large build (the repair path of tests/test_gpu_nan.py inside the 1024-thread blocks)
    x1[1000, dim - 1] = np.nan
    x2[2000, 0] = np.nan
    with np.errstate(all="ignore"):
        wn = osph.sphere_gaussian_kernel(x1, x2, beta)
    gn = ops.sphere_pairwise(t(x1), t(x2), beta=beta).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(gn), np.isnan(wn))
    np.testing.assert_allclose(gn[~np.isnan(wn)], wn[~np.isnan(wn)], rtol=5e-12, atol=1e-300)


def test_sphere_gram_with_a_cached_kernel_value_table_is_bit_identical():
    """gabo_sphere_pairwise_cached with a table from gabo_sphere_ktable_build against the launch that builds the table itself (ktable = NULL):
    the same bits; the ops layer keeps one table per (device, stream) and rebuilds it when beta changes."""
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, dim = 2304, 10
    x = rng.standard_normal((n, dim))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    X = t(x)
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.gabo_sphere_pairwise_uses_ktable(1, n, n, dim, 1.29, _lib.GABO_OUT_GAUSSIAN, 0) == 1
    assert lib.gabo_sphere_pairwise_uses_ktable(1, 100, 100, dim, 1.29, _lib.GABO_OUT_GAUSSIAN, 0) == 0
    assert lib.gabo_sphere_pairwise_uses_ktable(1, n, n, dim, 1.29, _lib.GABO_OUT_LAPLACE, 0) == 0
    table = torch.empty(int(lib.gabo_sphere_ktable_doubles()), dtype=torch.float64, device="cuda:0")
    outs = []
    for beta in (1.29, 0.4):
        _lib.check(lib.gabo_sphere_ktable_build(beta, table.data_ptr(), stream), "build")
        a, b = torch.empty(n, n, dtype=torch.float64, device="cuda:0"), torch.empty(n, n, dtype=torch.float64, device="cuda:0")
        _lib.check(lib.gabo_sphere_pairwise_cached(X.data_ptr(), X.data_ptr(), a.data_ptr(), 1, n, n, dim, 0, 0, beta, _lib.GABO_OUT_GAUSSIAN, 0, table.data_ptr(), stream), "cached")
        _lib.check(lib.gabo_sphere_pairwise(X.data_ptr(), X.data_ptr(), b.data_ptr(), 1, n, n, dim, 0, 0, beta, _lib.GABO_OUT_GAUSSIAN, 0, stream), "plain")
        assert torch.equal(a, b)
        outs.append(a.cpu().numpy())
        np.testing.assert_allclose(outs[-1], osph.sphere_gaussian_kernel(x, x, beta), rtol=5e-12, atol=1e-300)
    # the ops layer: the cache follows beta
    for beta, want in zip((1.29, 0.4, 1.29), (outs[0], outs[1], outs[0])):
        np.testing.assert_array_equal(ops.sphere_pairwise(X, X, beta=beta).cpu().numpy(), want)
    assert lib.gabo_sphere_ktable_build(9.0, table.data_ptr(), stream) != 0       # beyond the table's range of beta
