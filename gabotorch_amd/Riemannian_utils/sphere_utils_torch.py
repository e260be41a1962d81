"""Torch-facing sphere utilities with the reference's names (BoManifolds/Riemannian_utils/sphere_utils_torch.py)."""
import torch

from .. import _lib, ops


def sphere_distance_torch(x1, x2, diag=False):
    """acos(clamp(<x1_i, x2_j>))   (sphere_utils_torch.py:12-55)."""
    return ops.sphere_kernel(x1, x2, 1.0, _lib.GABO_OUT_DISTANCE, diag=diag)


def _geodesic_rotation_frame(x, y):
    """The rotation that carries the unit vector x to the unit vector y along their geodesic acts in the plane span{x, y} only.  With
    t = <x, y> (clamped like the reference, sphere_utils_torch.py:78-80), s = sqrt(1 - t^2) and the unit vector e orthogonal to y in
    that plane, e = (x - t y) / |x - t y|, it is
        R = I + s (y e^T - e y^T) + (t - 1) (y y^T + e e^T)            (Jung, Dryden & Marron 2012, appendix),
    a rank-2 update of the identity.  Returns (y, e, s, t) with x, y flattened to vectors."""
    x, y = x.reshape(-1), y.reshape(-1)
    t = torch.dot(x, y).clamp(-1.0 + 1e-15, 1.0 - 1e-15)
    e = x - t * y
    e = e / torch.linalg.vector_norm(e)
    s = torch.sqrt((1.0 - t) * (1.0 + t))          # sin(acos t) without the round trip through the angle
    return y, e, s, t


def rotate_along_geodesic(points, x, y, inverse=False):
    """rows of `points` (..., d) multiplied by R(x -> y)^T (i.e. every point rotated by R), or by R (inverse=True: rotated back),
    WITHOUT forming the d x d matrix: two inner products per point and a rank-2 correction.  Differentiable in points, x and y."""
    yv, e, s, t = _geodesic_rotation_frame(x.to(points), y.to(points))
    a = points @ yv                                  # <p, y>
    b = points @ e                                   # <p, e>
    if inverse:
        s = -s
    cy = s * b + (t - 1.0) * a                       # R p = p + (s <p,e> + (t-1) <p,y>) y + (-s <p,y> + (t-1) <p,e>) e
    ce = -s * a + (t - 1.0) * b
    return points + cy.unsqueeze(-1) * yv + ce.unsqueeze(-1) * e


def rotation_from_sphere_points_torch(x, y):
    """The rotation matrix moving x to y along the geodesic (sphere_utils_torch.py:58-93), for callers that want the matrix itself:
    the rank-2 form applied to the identity."""
    d = x.reshape(-1).shape[0]
    eye = torch.eye(d, dtype=x.dtype, device=x.device)
    return rotate_along_geodesic(eye, x, y).transpose(-1, -2)      # rows of the result of rotating e_k are the columns of R
