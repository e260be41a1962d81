"""Oracle (CPU, fp64, numpy) for the sphere half of the hot path.  Test infrastructure only - see oracle/__init__.py."""
import numpy as np

CLAMP = 1e-15       # sphere_utils_torch.py:53  clamp(-1 + 1e-15, 1 - 1e-15)


def sphere_distance(x1, x2, diag=False):
    """acos(clamp(<x1_i, x2_j>))   (Riemannian_utils/sphere_utils_torch.py:12-55).

    x1 (..., N1, dim), x2 (..., N2, dim) -> (..., N1, N2); diag=True pairs row k with row k -> (N, 1)
    (the reference's diag branch is 2-D only, :45-49)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    if diag:
        ip = np.sum(x1 * x2, axis=-1, keepdims=True)
    else:
        ip = np.einsum("...id,...jd->...ij", x1, x2)
    return np.arccos(np.clip(ip, -1.0 + CLAMP, 1.0 - CLAMP))


def sphere_gaussian_kernel(x1, x2, beta, diag=False):
    """exp(-beta d^2)   (kernel_utils/kernels_sphere.py:71-94)."""
    d = sphere_distance(x1, x2, diag)
    return np.exp(-(d * d) * beta)


def sphere_laplace_kernel(x1, x2, beta):
    """exp(-beta d)   (kernels_sphere.py:97-134)."""
    return np.exp(-sphere_distance(x1, x2) * beta)


def sphere_gaussian_kernel_grads(x1, x2, beta, grad_k):
    """d/dx1, d/dx2 of sum(grad_k * K).  dd/dc = -1/sqrt(1-c^2); zero where the clamp is active (autograd `clamp`
    semantics: strict inside the interval passes the gradient)   (SURVEY App. C)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    ip = np.einsum("...id,...jd->...ij", x1, x2)
    lo, hi = -1.0 + CLAMP, 1.0 - CLAMP
    c = np.clip(ip, lo, hi)
    d = np.arccos(c)
    k = np.exp(-(d * d) * beta)
    inside = (ip >= lo) & (ip <= hi)
    dk_dc = np.where(inside, k * (-beta) * 2.0 * d * (-1.0 / np.sqrt(1.0 - c * c)), 0.0)
    w = np.asarray(grad_k) * dk_dc
    return np.einsum("...ij,...jd->...id", w, x2), np.einsum("...ij,...id->...jd", w, x1)


# ------------------------------------------------------------------ exp / log maps (reference numpy statements)
def expmap(u, x0):
    """x0 cos|u| + u sin|u| / |u|; returns x0 where |u| < 1e-16   (Riemannian_utils/sphere_utils.py:14-38).  (..., dim)."""
    u = np.asarray(u, dtype=np.float64)
    x0 = np.asarray(x0, dtype=np.float64)
    nu = np.sqrt(np.sum(u * u, axis=-1, keepdims=True))
    safe = np.where(nu < 1e-16, 1.0, nu)
    x = x0 * np.cos(nu) + u * np.sin(nu) / safe
    return np.where(nu < 1e-16, x0, x)


def logmap(x, x0):
    """(x - x0 cos t) t / sin t, t = acos(clip(<x0,x>, -1, 1)); zero where t < 1e-16   (sphere_utils.py:41-65)."""
    x = np.asarray(x, dtype=np.float64)
    x0 = np.asarray(x0, dtype=np.float64)
    t = np.arccos(np.clip(np.sum(x * x0, axis=-1, keepdims=True), -1.0, 1.0))
    safe = np.where(t < 1e-16, 1.0, np.sin(t))
    u = (x - x0 * np.cos(t)) * t / safe
    return np.where(t < 1e-16, 0.0, u)


def rotation_from_sphere_points(x, y):
    """Rotation moving x to y along the geodesic   (sphere_utils_torch.py:58-93)."""
    x = np.asarray(x, dtype=np.float64).reshape(1, -1)
    y = np.asarray(y, dtype=np.float64).reshape(1, -1)
    dim = x.shape[1]
    ip = np.clip(x @ y.T, -1.0 + CLAMP, 1.0 - CLAMP)
    c = x - y * ip
    c = c / np.linalg.norm(c)
    return np.eye(dim) + np.sin(np.arccos(ip)) * (y.T @ c - c.T @ y) + (ip - 1.0) * (y.T @ y + c.T @ c)


# ------------------------------------------------------------------ pymanopt Sphere ops [3P; unpinned]
def proj(x, h):
    return h - np.sum(x * h, axis=-1, keepdims=True) * x


def retr(x, u):
    y = x + u
    return y / np.linalg.norm(y, axis=-1, keepdims=True)


def ehess2rhess(x, eg, eh, u):
    return proj(x, eh) - np.sum(x * eg, axis=-1, keepdims=True) * u


def transp(x, y, u):
    return proj(y, u)
