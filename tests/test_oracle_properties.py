"""Size-independent properties of the CPU oracle (hypothesis): the oracle is what the GPU path is judged against, so its own
invariants are checked independently of the golden vectors."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import spd as ospd
from oracle import sphere as osph


def _spd(rng, n, d):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.1, 4.0, (n, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), d=st.integers(2, 8))
def test_spd_distance_invariances(seed, d):
    rng = np.random.default_rng(seed)
    X, Y = _spd(rng, 4, d), _spd(rng, 5, d)
    D = ospd.affine_invariant_distance(X, Y)
    np.testing.assert_allclose(D, ospd.affine_invariant_distance(Y, X).T, rtol=1e-9, atol=1e-10)           # symmetry
    A = rng.standard_normal((d, d)) + 3 * np.eye(d)
    cong = lambda M: A @ M @ A.T                                                                           # noqa: E731
    np.testing.assert_allclose(ospd.affine_invariant_distance(cong(X), cong(Y)), D, rtol=1e-7, atol=1e-8)   # affine invariance
    np.testing.assert_allclose(ospd.affine_invariant_distance(np.linalg.inv(X), np.linalg.inv(Y)), D, rtol=1e-7, atol=1e-8)
    v = ospd.symmetric_matrix_to_vector_mandel(X)
    np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel(v), X, rtol=1e-14, atol=1e-15)        # Mandel round trip
    np.testing.assert_allclose((v * v).sum(-1), (X * X).sum((-1, -2)), rtol=1e-12)                          # Mandel is an isometry
    np.testing.assert_allclose(ospd.expmap(ospd.logmap(Y[:4], X), X), Y[:4], rtol=1e-8, atol=1e-9)          # exp o log = id
    U = ospd.logmap(Y[:4], X)
    np.testing.assert_allclose(np.sqrt(ospd.spd_inner(X, U, U) + 1e-15), np.diagonal(D[:, :4]), rtol=1e-8)  # |Log_X Y|_X = d(X,Y)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), dim=st.integers(2, 30))
def test_sphere_invariances(seed, dim):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((6, dim)); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.standard_normal((6, dim)); y /= np.linalg.norm(y, axis=1, keepdims=True)
    D = osph.sphere_distance(x, y)
    assert (D >= 0).all() and (D <= np.pi).all()
    np.testing.assert_allclose(D, osph.sphere_distance(y, x).T, rtol=1e-13)
    Q = np.linalg.qr(rng.standard_normal((dim, dim)))[0]
    np.testing.assert_allclose(osph.sphere_distance(x @ Q, y @ Q), D, rtol=1e-9, atol=1e-7)                 # rotation invariance
    u = osph.logmap(y, x)
    np.testing.assert_allclose(np.linalg.norm(u, axis=1), np.diagonal(D), rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(osph.expmap(u, x), y, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(np.sum(u * x, axis=1), 0.0, atol=1e-9)                                       # tangent


def test_gp_oracle_known_answers():
    from oracle import gp as ogp
    # noise-free GP interpolates its data with zero variance
    x = np.array([[0.0], [1.0], [2.5]])
    k = np.exp(-0.5 * (x - x.T) ** 2)
    y = np.array([1.0, -2.0, 0.5])
    mu, var = ogp.gp_posterior(k, k, np.ones(3), y, 0.2, 1.3, 1e-12)
    np.testing.assert_allclose(mu, y, atol=1e-8)
    np.testing.assert_allclose(var, 0.0, atol=1e-8)
    # far from the data the posterior is the prior
    mu, var = ogp.gp_posterior(k, np.zeros((1, 3)), np.ones(1), y, 0.2, 1.3, 1e-2)
    np.testing.assert_allclose([mu[0], var[0]], [0.2, 1.3], atol=1e-14)
    # EI: u = 0 gives sigma * phi(0); the variance clamp at 1e-9; symmetry between maximize and minimize
    np.testing.assert_allclose(ogp.expected_improvement(np.array([1.0]), np.array([4.0]), 1.0, True), 2.0 / np.sqrt(2 * np.pi))
    np.testing.assert_allclose(ogp.expected_improvement(np.array([1.0]), np.array([-1.0]), 1.0, True), np.sqrt(1e-9) / np.sqrt(2 * np.pi))
    np.testing.assert_allclose(ogp.expected_improvement(np.array([0.3]), np.array([0.5]), 1.0, False),
                               ogp.expected_improvement(np.array([1.7]), np.array([0.5]), 1.0, True))


def test_gp_mll_oracle_known_answers():
    """The marginal likelihood restatement against scipy's multivariate normal density and central differences of itself."""
    from scipy.stats import multivariate_normal
    from oracle import gp as ogp
    rng = np.random.default_rng(7)
    for n, p in ((1, 2), (6, 2), (23, 1)):
        pts = rng.standard_normal((n, 3))
        d = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
        e = d ** p
        y = rng.standard_normal(n)
        par = np.array([0.7, 1.9, 0.05, 0.3])
        ll, g = ogp.marginal_log_likelihood(e, y, *par)
        ky = par[1] * np.exp(-par[0] * e) + par[2] * np.eye(n)
        np.testing.assert_allclose(ll, multivariate_normal(mean=np.full(n, par[3]), cov=ky).logpdf(y), rtol=1e-11)
        for k in range(4):
            h = 1e-6 * max(1.0, abs(par[k]))
            up, dn = par.copy(), par.copy()
            up[k] += h
            dn[k] -= h
            fd = (ogp.marginal_log_likelihood(e, y, *up)[0] - ogp.marginal_log_likelihood(e, y, *dn)[0]) / (2 * h)
            np.testing.assert_allclose(g[k], fd, rtol=2e-6, atol=1e-7)


def test_faithful_mandel_loop_equals_the_vectorised_map():
    """the per-vector loop the cpu_baseline leg times (spd_utils_torch.py:172-194: it DIVIDES by 2**0.5) and the vectorised statement (which
    multiplies by the reciprocal): the same matrices to an ulp"""
    rng = np.random.default_rng(11)
    for d in (2, 5, 10):
        v = rng.standard_normal((7, d * (d + 1) // 2))
        np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel_faithful(v), ospd.vector_to_symmetric_matrix_mandel(v), rtol=3e-16, atol=0)
    vb = rng.standard_normal((2, 3, 6))
    assert ospd.vector_to_symmetric_matrix_mandel_faithful(vb).shape == (2, 3, 3, 3)
