"""Non-finite and non-SPD inputs: the HIP Gram kernels return what the reference returns on the same inputs.

Reference behaviour (probed by importing it, torch.symeig -> torch.linalg.eigh):
  * sphere: clamp -> acos -> exp propagate a NaN inner product (sphere_utils_torch.py:53-55, kernels_sphere.py:90-94): a NaN row of x1 gives a
    NaN row of K, a NaN row of x2 a NaN column, diag=True a NaN entry; an infinite inner product is clamped like any other (finite distance).
  * SPD: torch.cholesky(x1) raises on a NaN or non-positive-definite matrix of x1 (spd_utils_torch.py:87); NaN entries in x2 make the
    eigen-solver raise (RuntimeError as well); a matrix of x2 that is free of NaN but not positive definite goes through symeig and log and
    gives a NaN column, without an exception (spd_utils_torch.py:109-120).
The oracle (numpy) states the same arithmetic, so the comparison is entry by entry with NaN == NaN."""
import warnings

import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, ops
from oracle import spd as ospd
from oracle import sphere as osph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = {"gaussian": _lib.GABO_OUT_GAUSSIAN, "laplace": _lib.GABO_OUT_LAPLACE, "distance": _lib.GABO_OUT_DISTANCE}


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def _sphere_oracle(x1, x2, beta, mode, diag=False):
    with np.errstate(all="ignore"):
        d = osph.sphere_distance(x1, x2, diag=diag)
        if mode == "distance":
            return d
        return np.exp(-d * beta) if mode == "laplace" else np.exp(-(d * d) * beta)


def _sphere_points(rng, n, dim):
    x = rng.standard_normal((n, dim))
    return x / np.linalg.norm(x, axis=1, keepdims=True)


# dim 3 / 10 / 16: operands in registers (1, 3 and 4 MFMA steps, the padded one included); dim 21: the general path
@pytest.mark.parametrize("dim", [3, 10, 16, 21])
@pytest.mark.parametrize("mode,beta", [("gaussian", 1.3), ("gaussian", 2500.0), ("laplace", 0.7), ("distance", 1.0)])
@pytest.mark.parametrize("shape", [(70, 130), (256, 512)])
def test_sphere_gram_propagates_nan_like_the_reference(dim, mode, beta, shape):
    rng = np.random.default_rng(dim * 100 + shape[0])
    n1, n2 = shape
    x1, x2 = _sphere_points(rng, n1, dim), _sphere_points(rng, n2, dim)
    bad_rows, bad_cols = [1, n1 - 1, n1 // 2], [0, n2 // 3, n2 - 2]
    x1[bad_rows[0], dim - 1] = np.nan           # the LAST entry: the one the padded K lanes of the MFMA repeat
    x1[bad_rows[1], 0] = np.nan
    x1[bad_rows[2], :] = np.nan
    x2[bad_cols[0], dim - 1] = np.nan
    x2[bad_cols[1], dim // 2] = np.nan
    x2[bad_cols[2], :] = np.nan
    want = _sphere_oracle(x1, x2, beta, mode)
    got = ops.sphere_pairwise(t(x1), t(x2), beta=beta, mode=MODES[mode]).cpu().numpy()
    nan_want = np.isnan(want)
    assert nan_want[bad_rows].all() and nan_want[:, bad_cols].all() and nan_want.sum() == 3 * n2 + 3 * n1 - 9
    np.testing.assert_array_equal(np.isnan(got), nan_want)
    np.testing.assert_allclose(got[~nan_want], want[~nan_want], rtol=1e-12, atol=1e-13)
    # the clean rows / columns are bit-identical to a run without the corrupted points (the repair only touches NaN entries)
    keep_r = np.setdiff1d(np.arange(n1), bad_rows)
    keep_c = np.setdiff1d(np.arange(n2), bad_cols)
    clean = ops.sphere_pairwise(t(x1[keep_r]), t(x2[keep_c]), beta=beta, mode=MODES[mode]).cpu().numpy()
    np.testing.assert_array_equal(got[np.ix_(keep_r, keep_c)], clean)


@pytest.mark.parametrize("dim", [3, 10, 21])
def test_sphere_gram_nan_in_the_symmetric_build_and_in_a_batch(dim):
    rng = np.random.default_rng(7 + dim)
    x = _sphere_points(rng, 200, dim)
    x[17, dim - 1] = np.nan
    x[150, 0] = np.nan
    want = _sphere_oracle(x, x, 0.9, "gaussian")
    got = ops.sphere_pairwise(t(x), t(x), beta=0.9, symmetric=True).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-12, atol=1e-13)
    xb = np.stack([_sphere_points(rng, 40, dim) for _ in range(3)])
    xb[1, 5, dim - 1] = np.nan
    wantb = np.stack([_sphere_oracle(xb[k], xb[k], 1.1, "gaussian") for k in range(3)])
    gotb = ops.sphere_pairwise(t(xb), t(xb), beta=1.1).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(gotb), np.isnan(wantb))
    assert not np.isnan(gotb[0]).any() and not np.isnan(gotb[2]).any()


@pytest.mark.parametrize("mode,beta", [("gaussian", 1.3), ("laplace", 0.7), ("distance", 1.0)])
def test_sphere_diag_propagates_nan(mode, beta):
    rng = np.random.default_rng(3)
    x1, x2 = _sphere_points(rng, 50, 10), _sphere_points(rng, 50, 10)
    x1[4, 9] = np.nan
    x2[30, 0] = np.nan
    want = _sphere_oracle(x1, x2, beta, mode, diag=True)
    got = ops.sphere_pairwise(t(x1), t(x2), beta=beta, mode=MODES[mode], diag=True).cpu().numpy()
    assert np.isnan(want).sum() == 2
    np.testing.assert_allclose(got, want.reshape(got.shape), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("dim", [3, 10, 14, 21])
def test_sphere_gram_infinite_and_huge_entries(dim):
    """An infinite inner product is clamped (finite distance), inf - inf and inf * 0 are NaN: the numpy oracle on the same inputs decides,
    entry by entry."""
    rng = np.random.default_rng(dim)
    x1, x2 = _sphere_points(rng, 96, dim), _sphere_points(rng, 160, dim)
    x1[3, dim - 1] = np.inf                # last entry (padded MFMA lanes)
    x1[40, 0] = -np.inf
    x2[7, dim - 1] = np.inf
    x2[100, 1] = np.inf
    x2[101, 1] = np.inf
    x2[101, 2] = -np.inf
    x1[60, :] = 0.0
    x1[60, 1] = 1.0                        # zeros against infinities
    # (finite operands whose PRODUCTS overflow with both signs are left out: a fused multiply-add chain - the MFMA here, the FMA kernels of the
    # reference's BLAS - adds the unrounded product and stays at +-inf where separate multiplications and additions give inf - inf = NaN, so
    # the reference's own result depends on its BLAS build)
    x1[70, :] = 1e200
    x2[120, :] = 1e-200                    # huge against tiny: finite products
    for mode, beta in (("gaussian", 1.3), ("distance", 1.0)):
        want = _sphere_oracle(x1, x2, beta, mode)
        got = ops.sphere_pairwise(t(x1), t(x2), beta=beta, mode=MODES[mode]).cpu().numpy()
        # (the summation order of the MFMA and of numpy's dot product may differ where +inf and -inf meet a finite partial sum: both NaN)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-12, atol=1e-13)
    assert np.isnan(want).any() and np.isfinite(want[3]).any() and np.isfinite(want[:, 7]).any()


def _rand_spd(rng, n, d, lo=0.2, hi=4.0):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def _spd_oracle(m1, m2, beta, mode):
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = ospd.affine_invariant_distance(m1, m2)
        if mode == "distance":
            return d
        return np.exp(-d * beta) if mode == "laplace" else np.exp(-(d * d) * beta)


@pytest.mark.parametrize("d", [2, 3, 5, 10, 13])
@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
def test_spd_gram_nan_input_raises_like_the_reference(d, mode, raising):
    rng = np.random.default_rng(d)
    m1, m2 = _rand_spd(rng, 40, d), _rand_spd(rng, 70, d)
    v1, v2 = ospd.symmetric_matrix_to_vector_mandel(m1), ospd.symmetric_matrix_to_vector_mandel(m2)
    ops.spd_ai_pairwise(t(v1), t(v2), beta=0.8, mode=MODES[mode])                  # clean inputs: no exception
    for which in (1, 2):
        for pos in (0, d, v1.shape[1] - 1):                                        # a diagonal entry, an off-diagonal entry, the last entry
            a, b = v1.copy(), v2.copy()
            (a if which == 1 else b)[11, pos] = np.nan
            with raising("not positive definite"):
                ops.spd_ai_pairwise(t(a), t(b), beta=0.8, mode=MODES[mode])
    # a matrix of x1 that is not positive definite: torch.cholesky raises in the reference (spd_utils_torch.py:87)
    a = v1.copy()
    a[5, 0] = -1.0
    with raising("gabo_spd_ai_pairwise: input matrix #5 is not positive definite"):
        ops.spd_ai_pairwise(t(a), t(v2), beta=0.8, mode=MODES[mode])
    ops.check_deferred()                                                            # nothing is left pending


@pytest.mark.parametrize("d", [2, 3, 5, 10, 13])
@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
def test_spd_gram_indefinite_x2_gives_a_nan_column_like_the_reference(d, mode):
    rng = np.random.default_rng(10 + d)
    m1, m2 = _rand_spd(rng, 70, d), _rand_spd(rng, 130, d)
    # column 9: one negative eigenvalue; column 64: negative definite; column 129: the LAST pivot fails only
    w, q = np.linalg.eigh(m2[9])
    w[0] = -0.5
    m2[9] = (q * w) @ q.T
    m2[64] = -m2[64]
    m2[129] = np.eye(d)
    m2[129, d - 1, d - 1] = -1.0
    for k in (9, 64, 129):
        m2[k] = 0.5 * (m2[k] + m2[k].T)
    want = _spd_oracle(m1, m2, 0.8, mode)
    v1, v2 = ospd.symmetric_matrix_to_vector_mandel(m1), ospd.symmetric_matrix_to_vector_mandel(m2)
    got = ops.spd_ai_pairwise(t(v1), t(v2), beta=0.8, mode=MODES[mode]).cpu().numpy()
    nan_want = np.isnan(want)
    assert nan_want[:, [9, 64, 129]].all() and nan_want.sum() == 3 * 70
    np.testing.assert_array_equal(np.isnan(got), nan_want)
    np.testing.assert_allclose(got[~nan_want], want[~nan_want], rtol=1e-9, atol=1e-12)
    if mode == "gaussian":       # ... and with the distance matrix requested from the same launch
        k2, d2 = ops.spd_ai_pairwise(t(v1), t(v2), beta=0.8, return_dist=True)
        np.testing.assert_array_equal(np.isnan(k2.cpu().numpy()), nan_want)
        np.testing.assert_array_equal(np.isnan(d2.cpu().numpy()), nan_want)


@pytest.mark.parametrize("mode", ["gaussian", "laplace", "distance"])
def test_frobenius_and_log_euclidean_gram_propagate_nan(mode):
    """exp(-beta |x - y|^2) of a NaN difference is NaN in the reference (torch.exp, kernels_spd.py:238-240, 309-311); the table exp of the
    Frobenius epilogue clamps its argument and used to return 0.  A NaN row of x1 -> NaN row, of x2 -> NaN column; a matrix with a negative
    eigenvalue has a NaN logm (spd_utils_torch.py:13-30: torch.log of the eigenvalues), hence a NaN row / column of the log-Euclidean Gram."""
    rng = np.random.default_rng(77)
    d = 3
    m1, m2 = _rand_spd(rng, 70, d), _rand_spd(rng, 130, d)
    v1, v2 = ospd.symmetric_matrix_to_vector_mandel(m1), ospd.symmetric_matrix_to_vector_mandel(m2)
    a, b = v1.copy(), v2.copy()
    a[7, 2] = np.nan
    b[64, 0] = np.nan
    got = ops.frobenius_pairwise(t(a), t(b), beta=0.6, mode=MODES[mode]).cpu().numpy()
    with np.errstate(all="ignore"):
        dist = ospd.frobenius_distance(ospd.vector_to_symmetric_matrix_mandel(a), ospd.vector_to_symmetric_matrix_mandel(b))
        want = dist if mode == "distance" else (np.exp(-dist * 0.6) if mode == "laplace" else np.exp(-(dist * dist) * 0.6))
    nan_want = np.isnan(want)
    assert nan_want[7].all() and nan_want[:, 64].all() and nan_want.sum() == 130 + 70 - 1
    np.testing.assert_array_equal(np.isnan(got), nan_want)
    np.testing.assert_allclose(got[~nan_want], want[~nan_want], rtol=1e-9, atol=1e-12)
    # log-Euclidean: logm of an indefinite matrix is NaN, and so is every kernel value it enters
    w, q = np.linalg.eigh(m2[9])
    w[0] = -0.5
    m2b = m2.copy()
    m2b[9] = (q * w) @ q.T
    m2b[9] = 0.5 * (m2b[9] + m2b[9].T)
    l1 = ops.spd_logm_mandel(t(v1))
    l2 = ops.spd_logm_mandel(t(ospd.symmetric_matrix_to_vector_mandel(m2b)))
    assert bool(torch.isnan(l2[9]).any()) and not bool(torch.isnan(l2[:9]).any())
    k = ops.frobenius_pairwise(l1, l2, beta=0.6, mode=MODES[mode]).cpu().numpy()
    assert np.isnan(k[:, 9]).all() and np.isnan(k).sum() == 70
