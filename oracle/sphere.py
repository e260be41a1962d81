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


def sphere_gaussian_kernel_hvp(x1, x2, beta, grad_k, u):
    """Second order: d/dt grad_x1 [sum(grad_k * K(x1 + t u, x2))] at t = 0 for the Gaussian kernel, Euclidean coordinates (what the reference's
    PyTorch backend returns as `ehess`, pymanopt_addons/tools/autodiff/_pytorch.py:103-116, through sphere_utils_torch.py:12-55 and
    kernels_sphere.py:90-94).  K = phi(c), c = <x1_i, x2_j>:  Hess[U]_i = sum_j grad_k_ij phi''(c_ij) <x2_j, u_i> x2_j with
    phi' = 2 beta psi phi,  phi'' = 2 beta phi (2 beta psi^2 - psi' / sin(theta)),  theta = acos c,  psi = theta / sin(theta),
    psi' = (sin(theta) - theta cos(theta)) / sin^2(theta); zero where the clamp is active.  Pinned by tests/golden/hvp.npz."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    ip = np.einsum("...id,...jd->...ij", x1, x2)
    lo, hi = -1.0 + CLAMP, 1.0 - CLAMP
    c = np.clip(ip, lo, hi)
    th = np.arccos(c)
    sn = np.sqrt(1.0 - c * c)
    phi = np.exp(-(th * th) * beta)
    psi = th / sn
    dpsi = (sn - th * c) / (sn * sn)
    d2 = 2.0 * beta * phi * (2.0 * beta * psi * psi - dpsi / sn)
    d2 = np.where((ip >= lo) & (ip <= hi), d2, 0.0)
    w = np.asarray(grad_k) * d2 * np.einsum("...jd,...id->...ij", x2, np.asarray(u, dtype=np.float64))
    return np.einsum("...ij,...jd->...id", w, x2)


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


# ------------------------------------------------------------------------------- nested spheres (principal nested spheres)
def _north(dim):
    n = np.zeros((1, dim))
    n[0, -1] = 1.0
    return n


def projection_from_sphere_to_nested_sphere(x, axis, dist_to_axis):
    """Projection of points of S^d onto the small circle {distance to `axis` = dist_to_axis}, computed after rotating the
    axis to the north pole and rotating back   (nested_spheres_utils.py:13-65; note the + 1e-6 in the rescaling :56-57)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, np.shape(x)[-1])
    dim = x.shape[-1]
    north = _north(dim)
    rot = rotation_from_sphere_points(np.asarray(axis).reshape(1, -1), north)
    xr = x @ rot.T
    theta = sphere_distance(xr, north)                                        # (N, 1)
    r = float(np.asarray(dist_to_axis).reshape(-1)[0])
    y = (np.sin(r) * xr + np.sin(theta - r) * north) / (np.sin(theta) + 1e-6)
    return y @ rot


def projection_from_sphere_to_next_subsphere(x, axis, dist_to_axis):
    """S^d -> S^(d-1): nested-sphere projection, rotation of the axis to the north pole, drop the last coordinate, rescale by
    1/(sin r + 1e-6) and renormalise with another + 1e-6   (nested_spheres_utils.py:68-114)."""
    shape = np.shape(x)
    x = np.asarray(x, dtype=np.float64).reshape(-1, shape[-1])
    dim = x.shape[-1]
    rot = rotation_from_sphere_points(np.asarray(axis).reshape(1, -1), _north(dim))
    r = float(np.asarray(dist_to_axis).reshape(-1)[0])
    y = projection_from_sphere_to_nested_sphere(x, axis, dist_to_axis)
    z = (y @ rot[:-1, :].T) / (np.sin(r) + 1e-6)
    z = z / (np.linalg.norm(z, axis=-1, keepdims=True) + 1e-6)
    return z.reshape(shape[:-1] + (dim - 1,))


def projection_from_sphere_to_subsphere(x, axes, dists):
    """[x, x_{d-1}, ..., x_{d-r}]   (nested_spheres_utils.py:117-146)."""
    out = [np.asarray(x, dtype=np.float64)]
    for a, r in zip(axes, dists):
        out.append(projection_from_sphere_to_next_subsphere(out[-1], a, r))
    return out


def projection_from_subsphere_to_next_sphere(z, axis, dist_to_axis):
    """S^(d-1) -> S^d: [sin r z, cos r] rotated from the north pole to the axis   (nested_spheres_utils.py:149-179)."""
    z = np.asarray(z, dtype=np.float64)
    dim = z.shape[-1] + 1
    rot = rotation_from_sphere_points(_north(dim), np.asarray(axis).reshape(1, -1))
    r = float(np.asarray(dist_to_axis).reshape(-1)[0])
    lifted = np.concatenate([np.sin(r) * z, np.cos(r) * np.ones(z.shape[:-1] + (1,))], axis=-1)
    return lifted @ rot.T


def projection_from_subsphere_to_sphere(z, axes, dists):
    """[z, x_{d-r+1}, ..., x_d], axes used in reverse order   (nested_spheres_utils.py:182-218)."""
    out = [np.asarray(z, dtype=np.float64)]
    for a, r in zip(list(axes)[::-1], list(dists)[::-1]):
        out.append(projection_from_subsphere_to_next_sphere(out[-1], a, r))
    return out


def nested_sphere_gaussian_kernel(x1, x2, axes, dists, beta):
    """exp(-beta d(p(x1), p(x2))^2)   (kernels_nested_sphere.py:125-152)."""
    p1 = projection_from_sphere_to_subsphere(x1, axes, dists)[-1]
    p2 = projection_from_sphere_to_subsphere(x2, axes, dists)[-1]
    d = sphere_distance(p1, p2)
    return np.exp(-beta * d * d)


def nested_sphere_reconstruction_cost(x, x_sub, axes, dists):
    """sum_n d(x_n, reconstruction_n)^2   (nested_spheres_optimization.py:20-38)."""
    rec = projection_from_subsphere_to_sphere(x_sub, axes, dists)[-1]
    d = sphere_distance(np.asarray(x, dtype=np.float64), rec, diag=True)
    return float(np.sum(d * d))
