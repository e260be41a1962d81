"""Oracle (CPU, fp64, numpy) for the SPD half of the hot path.  Test infrastructure only - see oracle/__init__.py.

All citations are paths under the reference tree (BoManifolds/...).  Everything computes in float64; the
reference's fp32 eigenvalue sink (Riemannian_utils/spd_utils_torch.py:108) is deliberately NOT reproduced,
so agreement with reference-generated vectors is ~1e-7 on distances while agreement with the independent
numpy statement of the distance (Riemannian_utils/spd_utils.py:180-197) is ~1e-13.
"""
import numpy as np

SQRT2 = 2.0 ** 0.5


# ----------------------------------------------------------------------------------------------- Mandel maps
def mandel_dim(d_vec):
    """d from d_vec = d(d+1)/2   (spd_utils_torch.py:175)."""
    d = int((-1.0 + (1.0 + 8.0 * d_vec) ** 0.5) / 2.0)
    if d * (d + 1) // 2 != d_vec:
        raise ValueError(f"{d_vec} is not a triangular number")
    return d


def mandel_index(d):
    """Row/col of every Mandel entry: the main diagonal first, then super-diagonal 1, 2, ... (spd_utils_torch.py:183-187)."""
    rows, cols = [], []
    for k in range(d):
        for i in range(d - k):
            rows.append(i)
            cols.append(i + k)
    return np.array(rows), np.array(cols)


def vector_to_symmetric_matrix_mandel(v):
    """(..., d_vec) -> (..., d, d); off-diagonals divided by sqrt(2)   (spd_utils_torch.py:159-194)."""
    v = np.asarray(v, dtype=np.float64)
    d = mandel_dim(v.shape[-1])
    r, c = mandel_index(d)
    scale = np.where(r == c, 1.0, 1.0 / SQRT2)
    m = np.zeros(v.shape[:-1] + (d, d))
    m[..., r, c] = v * scale
    m[..., c, r] = v * scale
    return m


def vector_to_symmetric_matrix_mandel_faithful(v):
    """The same map with the reference's OP SEQUENCE (spd_utils_torch.py:172-194): a Python loop over the vectors, each matrix assembled
    from its diagonals (one diagonal-matrix construction and addition per off-diagonal, above and below).  Used by the `cpu_baseline` leg of
    bench.py only, so that the timed port spends its time where the reference does; same values as the vectorised
    statement above (tests/test_oracle_properties.py)."""
    import torch
    vectors = torch.as_tensor(np.asarray(v, dtype=np.float64)).reshape(-1, np.shape(v)[-1])
    d = mandel_dim(vectors.shape[1])
    bounds = np.cumsum(range(d, 0, -1))
    out = torch.zeros(vectors.shape[0], d, d, dtype=torch.float64)
    for n in range(vectors.shape[0]):
        row = vectors[n]
        mat = torch.diag(row[0:d])
        for i in range(d - 1):
            seg = row[int(bounds[i]):int(bounds[i + 1])]
            mat = mat + torch.diag(seg, i + 1) / SQRT2
            mat = mat + torch.diag(seg, -i - 1) / SQRT2
        out[n] = mat
    return out.numpy().reshape(tuple(np.shape(v)[:-1]) + (d, d))


def symmetric_matrix_to_vector_mandel(m):
    """(..., d, d) -> (..., d_vec); off-diagonals sqrt(2) * mean(upper, lower)   (spd_utils_torch.py:197-226, :219)."""
    m = np.asarray(m, dtype=np.float64)
    d = m.shape[-1]
    r, c = mandel_index(d)
    scale = np.where(r == c, 1.0, SQRT2)
    return 0.5 * (m[..., r, c] + m[..., c, r]) * scale


# ----------------------------------------------------------------------------------------- affine-invariant
def _chol_inv(x):
    """L^-1 with L = lower Cholesky factor   (spd_utils_torch.py:87-88)."""
    L = np.linalg.cholesky(x)
    eye = np.broadcast_to(np.eye(x.shape[-1]), x.shape)
    return np.linalg.solve(L, eye)


def congruence_matrices(x1, x2):
    """M_ij = L_i^-1 X2_j L_i^-T for every pair: (..., N1, N2, d, d)   (spd_utils_torch.py:102-103)."""
    li = _chol_inv(np.asarray(x1, dtype=np.float64))
    x2 = np.asarray(x2, dtype=np.float64)
    return np.einsum("...iab,...jbc,...idc->...ijad", li, x2, li, optimize=True)


def affine_invariant_distance(x1, x2, diagonal_distance=False):
    """d_ij = sqrt(sum_k log^2 lambda_k(M_ij) + 1e-15)   (spd_utils_torch.py:53-121).

    x1 (..., N1, d, d), x2 (..., N2, d, d) -> (..., N1, N2).  diagonal_distance=True returns zeros (..., N2, 1)
    exactly as the reference does (:72-75)."""
    if diagonal_distance:
        return np.zeros(tuple(np.shape(x2)[:-2]) + (1,))
    m = congruence_matrices(x1, x2)
    # symeig(upper=True) reads the upper triangle only (:110)
    lam = np.linalg.eigvalsh(m, UPLO="U")
    lg = np.log(lam)
    return np.sqrt(np.sum(lg * lg, axis=-1) + 1e-15)


def affine_invariant_distance_faithful(x1, x2):
    """Same result, computed with the reference's OP SEQUENCE on torch CPU: Cholesky, explicit inverse, two
    batched products over the materialised N1*N2 set, then ONE symmetric eigensolve PER PAIR in a Python loop
    (spd_utils_torch.py:87-110).  This is the `cpu_baseline` ("port") that bench.py times: it is what the
    reference spends its time on.  Only difference: eigenvalues stay fp64 (no fp32 sink)."""
    import torch
    x1 = torch.as_tensor(x1, dtype=torch.float64)
    x2 = torch.as_tensor(x2, dtype=torch.float64)
    d = x1.shape[-1]
    n1, n2 = x1.shape[-3], x2.shape[-3]
    li = torch.inverse(torch.linalg.cholesky(x1))
    li_rep = li.unsqueeze(-3).expand(*li.shape[:-3], n1, n2, d, d).reshape(-1, d, d)
    x2_rep = x2.unsqueeze(-4).expand(*x2.shape[:-3], n1, n2, d, d).reshape(-1, d, d)
    m = torch.bmm(torch.bmm(li_rep, x2_rep), li_rep.transpose(-2, -1))
    lam = torch.empty(m.shape[0], d, dtype=torch.float64)
    for p in range(m.shape[0]):
        lam[p] = torch.linalg.eigh(m[p], UPLO="U").eigenvalues
    lg = torch.log(lam)
    out = torch.sqrt((lg * lg).sum(-1) + 1e-15)
    return out.reshape(*x1.shape[:-3], n1, n2).numpy()


def spd_ai_gaussian_kernel(x1_mandel, x2_mandel, beta, diagonal_distance=False, faithful=False):
    """K = exp(-beta d^2) from Mandel inputs   (kernel_utils/kernels_spd.py:72-100)."""
    to_matrix = vector_to_symmetric_matrix_mandel_faithful if faithful else vector_to_symmetric_matrix_mandel
    m1 = to_matrix(x1_mandel)
    m2 = to_matrix(x2_mandel)
    if faithful and not diagonal_distance:
        dist = affine_invariant_distance_faithful(m1, m2)
    else:
        dist = affine_invariant_distance(m1, m2, diagonal_distance)
    return np.exp(-(dist * dist) * beta)


def spd_ai_laplace_kernel(x1_mandel, x2_mandel, beta):
    """K = exp(-beta d)   (kernels_spd.py:157-187)."""
    dist = affine_invariant_distance(vector_to_symmetric_matrix_mandel(x1_mandel),
                                     vector_to_symmetric_matrix_mandel(x2_mandel))
    return np.exp(-dist * beta)


def spd_ai_gaussian_kernel_grads(x1_mandel, x2_mandel, beta, grad_k):
    """Closed-form d/dx1 and d/dx2 (Mandel) of sum(grad_k * K)   (the reference obtains these by
    autograd through cholesky/inverse/bmm/symeig, spd_utils_torch.py:87-120).

    grad_A d^2 = -2 L^-T logm(M) L^-1 ; grad_B d^2 = +2 L^-T M^-1 logm(M) L^-1 (= the same formula with the roles
    of A and B exchanged).  dK/dd^2 = -beta K, and the 1e-15 under the sqrt drops out because K depends on d^2."""
    a = vector_to_symmetric_matrix_mandel(x1_mandel)
    b = vector_to_symmetric_matrix_mandel(x2_mandel)
    li = _chol_inv(a)
    m = np.einsum("...iab,...jbc,...idc->...ijad", li, b, li, optimize=True)
    lam, v = np.linalg.eigh(m, UPLO="U")
    lg = np.log(lam)
    k = np.exp(-(np.sum(lg * lg, -1) + 1e-15) * beta)
    w = np.asarray(grad_k) * (-beta) * k                                   # dLoss/d(d^2_ij)
    logm = np.einsum("...ab,...b,...cb->...ac", v, lg, v)
    logm_minv = np.einsum("...ab,...b,...cb->...ac", v, lg / lam, v)
    ga = np.einsum("...ij,...iba,...ijbc,...icd->...iad", -2.0 * w, li, logm, li, optimize=True)
    gb = np.einsum("...ij,...iba,...ijbc,...icd->...jad", 2.0 * w, li, logm_minv, li, optimize=True)
    return symmetric_matrix_to_vector_mandel(ga), symmetric_matrix_to_vector_mandel(gb)


def _divided_differences(fun, dfun, lam):
    """first divided differences of `fun` at the eigenvalues, the derivative where two of them coincide (to rounding)"""
    a, b = lam[..., :, None], lam[..., None, :]
    close = np.abs(a - b) <= 1e-9 * (np.abs(a) + np.abs(b))
    den = np.where(close, 1.0, a - b)
    return np.where(close, dfun(0.5 * (a + b)), (fun(a) - fun(b)) / den)


def spd_ai_kernel_hvp(x1_mandel, x2_mandel, beta, grad_k, u_mandel, mode="gaussian"):
    """Second order: d/dt grad_x1 [sum(grad_k * K(x1 + t u, x2))] at t = 0, Mandel layout, with K the Gaussian / Laplace kernel or the
    distance itself.  The reference obtains it as torch.autograd.grad(<egrad, u>, x) through cholesky / inverse / bmm / symeig / log / exp
    (pymanopt_addons/tools/autodiff/_pytorch.py:103-116 `ehess`, Riemannian_utils/spd_utils_torch.py:87-120, kernels_spd.py:94-98, 185).

    Closed form per pair, A = x1_i = L L^T, M = L^-1 B L^-T = V diag(lam) V^T, f = sum log^2 lam, Ut = L^-1 U L^-T, U' = V^T Ut V:
        grad_A f      = -2 L^-T logm(M) L^-1,                <grad_A f, U> = -2 sum_k log(lam_k) U'_kk,
        Hess_A f [U]  =  2 L^-T V (C o U') V^T L^-1,          C_kl = divided differences of t log t  (C_kk = 1 + log lam_k),
        D^2 K [U]     =  phi'(f) Hess f [U] + phi''(f) <grad f, U> grad f,      K = phi(f)
    (phi(f) = exp(-beta (f + 1e-15)), exp(-beta sqrt(f + 1e-15)), sqrt(f + 1e-15)).  Pinned by tests/golden/hvp.npz, which is the
    reference's own `ehess`."""
    a = vector_to_symmetric_matrix_mandel(x1_mandel)
    b = vector_to_symmetric_matrix_mandel(x2_mandel)
    u = vector_to_symmetric_matrix_mandel(u_mandel)
    li = _chol_inv(a)
    m = np.einsum("...iab,...jbc,...idc->...ijad", li, b, li, optimize=True)
    lam, v = np.linalg.eigh(m, UPLO="U")
    lg = np.log(lam)
    f = np.sum(lg * lg, -1)
    s = f + 1e-15
    if mode == "gaussian":
        k = np.exp(-beta * s)
        p1, p2 = -beta * k, beta * beta * k
    elif mode == "laplace":
        r = np.sqrt(s)
        k = np.exp(-beta * r)
        p1, p2 = -beta * k / (2.0 * r), k * (beta * beta / (4.0 * s) + beta / (4.0 * s * r))
    elif mode == "distance":
        r = np.sqrt(s)
        p1, p2 = 1.0 / (2.0 * r), -1.0 / (4.0 * s * r)
    else:
        raise ValueError(mode)
    ut = np.einsum("...iab,...ibc,...idc->...iad", li, u, li, optimize=True)           # L^-1 U L^-T per row
    up = np.einsum("...ijba,...ibc,...ijcd->...ijad", v, ut, v, optimize=True)          # V^T Ut V per pair
    c = _divided_differences(lambda t: t * np.log(t), lambda t: 1.0 + np.log(t), lam)
    hess_f = 2.0 * np.einsum("...ab,...bc,...dc->...ad", v, c * up, v, optimize=True)    # in the whitened coordinates
    grad_f = -2.0 * np.einsum("...ab,...b,...cb->...ac", v, lg, v)
    inner = -2.0 * np.einsum("...k,...kk->...", lg, up)
    g = np.asarray(grad_k)
    core = (g * p1)[..., None, None] * hess_f + (g * p2 * inner)[..., None, None] * grad_f
    hv = np.einsum("...iba,...ijbc,...icd->...iad", li, core, li, optimize=True)
    return symmetric_matrix_to_vector_mandel(hv)


# ------------------------------------------------------------------------------------- matrix functions
def _sym_fun(x, f):
    lam, v = np.linalg.eigh(np.asarray(x, dtype=np.float64), UPLO="U")
    return np.einsum("...ab,...b,...cb->...ac", v, f(lam), v)


def logm(x):
    """V diag(log lambda) V^-1   (spd_utils_torch.py:13-30; tools/multi.py:55-64 multilog)."""
    return _sym_fun(x, np.log)


def sqrtm(x):
    """spd_utils_torch.py:33-50."""
    return _sym_fun(x, np.sqrt)


def expm_sym(x):
    """tools/multi.py:67-75 multiexp(sym=True)."""
    return _sym_fun(x, np.exp)


def frobenius_distance(x1, x2, diagonal_distance=False):
    """||x1_i - x2_j + 1e-15||_F with the 1e-15 added to EVERY element   (spd_utils_torch.py:124-156)."""
    if diagonal_distance:
        return np.zeros(tuple(np.shape(x2)[:-2]) + (1,))
    diff = np.asarray(x1)[..., :, None, :, :] - np.asarray(x2)[..., None, :, :, :] + 1e-15
    return np.sqrt(np.sum(diff * diff, axis=(-2, -1)))


def log_euclidean_distance(x1, x2):
    """frobenius_distance(logm(x1), logm(x2))   (kernels_spd.py:289-309)."""
    return frobenius_distance(logm(x1), logm(x2))


def projection_from_spd_to_nested_spd(x, w):
    """Y = W^T X W   (nested_mappings/nested_spd_utils.py:13-48)."""
    w = np.asarray(w, dtype=np.float64)
    return np.einsum("ba,...bc,cd->...ad", w, np.asarray(x, dtype=np.float64), w, optimize=True)


# --------------------------------------------------------------------------- exp / log maps & manifold ops
def expmap(u, s):
    """Exp_S(U) = S expm(S^-1 U); reference argument order (tangent first)   (spd_utils.py:104-120)."""
    s = np.asarray(s, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    li = _chol_inv(s)
    L = np.linalg.cholesky(s)
    inner = expm_sym(li @ u @ np.swapaxes(li, -1, -2))
    return L @ inner @ np.swapaxes(L, -1, -2)


def logmap(x, s):
    """Log_S(X) = S logm(S^-1 X); reference argument order (point first, base second)   (spd_utils.py:123-139)."""
    s = np.asarray(s, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    li = _chol_inv(s)
    L = np.linalg.cholesky(s)
    inner = logm(li @ x @ np.swapaxes(li, -1, -2))
    return L @ inner @ np.swapaxes(L, -1, -2)


def _sym(a):
    return 0.5 * (a + np.swapaxes(a, -1, -2))


def spd_inner(x, u, v):
    """tr(X^-1 U X^-1 V)   [3P pymanopt PositiveDefinite.inner; unpinned]."""
    xu = np.linalg.solve(x, u)
    xv = np.linalg.solve(x, v)
    return np.einsum("...ab,...ba->...", xu, xv)


def spd_norm(x, u):
    return np.sqrt(np.maximum(spd_inner(x, u, u), 0.0))


def spd_egrad2rgrad(x, g):
    """X sym(G) X   [3P]."""
    return x @ _sym(g) @ x


def spd_ehess2rhess(x, eg, eh, u):
    """X sym(eh) X + sym(U sym(eg) X)   [3P]."""
    return x @ _sym(eh) @ x + _sym(u @ _sym(eg) @ x)


def spd_exp(x, u):
    """pymanopt argument order (base first) - same map as `expmap(u, x)`."""
    return expmap(u, x)


def spd_log(x, y):
    """pymanopt argument order: Log at base x of y."""
    return logmap(y, x)


def spd_transp(x1, x2, d):
    """PositiveDefinite.transp is the identity   [3P]."""
    return d


def spd_sample(n, min_eig, max_eig):
    """Random SPD matrix drawn from numpy's GLOBAL RNG in the reference's draw order: rand(n) for the eigenvalues,
    then randn(n, n) for the orthogonal factor   (spd_utils.py:290-306)."""
    lam = min_eig * np.ones(1) + (max_eig - min_eig) * np.random.rand(n)
    q, _ = np.linalg.qr(np.random.randn(n, n))
    return q @ np.diag(lam) @ q.T


def max_eigenvalue_constraint(x, maximum_eigenvalue):
    """maximum_eigenvalue - lambda_max(x) and its Euclidean gradient -v_max v_max^T   (spd_constraints_utils_torch.py:17-32)."""
    lam, v = np.linalg.eigh(np.asarray(x, dtype=np.float64), UPLO="U")
    vm = v[..., :, -1]
    return maximum_eigenvalue - lam[..., -1], -vm[..., :, None] * vm[..., None, :]


def min_eigenvalue_constraint(x, minimum_eigenvalue):
    """lambda_min(x) - minimum_eigenvalue and gradient v_min v_min^T   (spd_constraints_utils_torch.py:35-50)."""
    lam, v = np.linalg.eigh(np.asarray(x, dtype=np.float64), UPLO="U")
    vm = v[..., :, 0]
    return lam[..., 0] - minimum_eigenvalue, vm[..., :, None] * vm[..., None, :]


# ------------------------------------------------------------------------- log-Euclidean kernel and its gradients
def dlogm_adjoint(a, g):
    """Adjoint of the Frechet derivative of the matrix logarithm at SPD `a`, applied to symmetric `g` (Daleckii-Krein):
    V ((V^T G V) o F) V^T with F_kl = (log l_k - log l_l)/(l_k - l_l), F_kk = 1/l_k.  This is what autograd through
    logm_torch (spd_utils_torch.py:13-30) computes, without the 1/(l_k - l_l) blow-up for (nearly) repeated eigenvalues."""
    lam, v = np.linalg.eigh(np.asarray(a, dtype=np.float64), UPLO="U")
    lg = np.log(lam)
    dl = lam[..., :, None] - lam[..., None, :]
    dlg = lg[..., :, None] - lg[..., None, :]
    mean = 0.5 * (lam[..., :, None] + lam[..., None, :])
    close = np.abs(dl) <= 1e-9 * mean
    f = np.where(close, 1.0 / mean, dlg / np.where(close, 1.0, dl))
    inner = np.swapaxes(v, -1, -2) @ np.asarray(g, dtype=np.float64) @ v
    return v @ (inner * f) @ np.swapaxes(v, -1, -2)


def log_euclidean_gaussian_kernel(x1_mandel, x2_mandel, lengthscale):
    """exp(-||logm X1_i - logm X2_j + 1e-15||_F^2 / lengthscale^2)   (kernels_spd.py:267-313)."""
    d = log_euclidean_distance(vector_to_symmetric_matrix_mandel(x1_mandel), vector_to_symmetric_matrix_mandel(x2_mandel))
    return np.exp(-(d * d) / (lengthscale * lengthscale))


def log_euclidean_gaussian_kernel_grads(x1_mandel, x2_mandel, lengthscale, grad_k):
    """d/dx1, d/dx2 (Mandel) of sum(grad_k * K_logEuclid)."""
    a = vector_to_symmetric_matrix_mandel(x1_mandel)
    b = vector_to_symmetric_matrix_mandel(x2_mandel)
    la, lb = logm(a), logm(b)
    diff = la[..., :, None, :, :] - lb[..., None, :, :, :] + 1e-15
    d2 = np.sum(diff * diff, axis=(-2, -1))
    k = np.exp(-d2 / lengthscale ** 2)
    w = np.asarray(grad_k) * k * (-2.0 / lengthscale ** 2)
    ga_log = np.einsum("...ij,...ijab->...iab", w, diff)
    gb_log = -np.einsum("...ij,...ijab->...jab", w, diff)
    return (symmetric_matrix_to_vector_mandel(dlogm_adjoint(a, _sym(ga_log))),
            symmetric_matrix_to_vector_mandel(dlogm_adjoint(b, _sym(gb_log))))


# ------------------------------------------------------------------------------- nested SPD reconstruction (f4)
def projection_from_nested_spd_to_spd(y, w, v, c, k):
    """X = R [[Y, B], [B^T, C]] R^T with R = [W, V], B = Y^1/2 K C^1/2   (nested_spd_utils.py:51-118)."""
    y = np.asarray(y, dtype=np.float64)
    single = y.ndim == 2
    if single:
        y = y[None]
    rot = np.concatenate([w, v], axis=1)
    sc = sqrtm(c)
    out = []
    for yn in y:
        side = sqrtm(yn) @ k @ sc
        xr = np.block([[yn, side], [side.T, c]])
        out.append(rot @ xr @ rot.T)
    out = np.stack(out)
    return out[0] if single else out


def reconstruction_cost(x, y, w, v, c, k, metric="ai"):
    """sum_n dist(X_n, reconstruction(Y_n))^2 with the affine-invariant ("ai") or log-Euclidean ("le") distance
    (nested_spd_optimization.py:23-92; the reference accumulates the N distances in a float32 tensor)."""
    xr = projection_from_nested_spd_to_spd(y, w, v, c, k)
    total = 0.0
    for xn, rn in zip(np.asarray(x, dtype=np.float64), xr):
        if metric == "ai":
            dist = affine_invariant_distance(xn[None], rn[None])[0, 0]
        else:
            dist = frobenius_distance(logm(xn)[None], logm(rn)[None])[0, 0]
        total += dist * dist
    return total


def matfun_adjoint(a, g, fn):
    """Adjoint of the Frechet derivative of fn in {"log", "sqrt"} at SPD a applied to g (what autograd through
    logm_torch / sqrtm_torch computes, spd_utils_torch.py:13-50)."""
    lam, v = np.linalg.eigh(np.asarray(a, dtype=np.float64), UPLO="U")
    f = np.log(lam) if fn == "log" else np.sqrt(lam)
    fp = 1.0 / lam if fn == "log" else 0.5 / np.sqrt(lam)
    dl = lam[..., :, None] - lam[..., None, :]
    df = f[..., :, None] - f[..., None, :]
    close = np.abs(dl) <= 1e-9 * np.abs(lam[..., :, None] + lam[..., None, :])
    dd = np.where(close, 0.5 * (fp[..., :, None] + fp[..., None, :]), df / np.where(close, 1.0, dl))
    gs = 0.5 * (np.asarray(g) + np.swapaxes(np.asarray(g), -1, -2))
    inner = np.swapaxes(v, -1, -2) @ gs @ v
    return v @ (inner * dd) @ np.swapaxes(v, -1, -2)
