"""Nested-sphere projections with the reference's names and signatures (BoManifolds/nested_mappings/nested_spheres_utils.py).

Per level S^d -> S^(d-1): the rotation of the axis to the north pole acts in one plane only, so it is applied to the points as a rank-2
update (two inner products per point, `rotate_along_geodesic`; the d x d matrix is never formed) and the point-wise part (distance to the axis, rescaling, renormalisation with the
reference's + 1e-6 terms) is the HIP epilogue gabo_nested_sphere_epilogue, differentiable through its HIP backward.  The
axes stay differentiable through the rotation matrix (torch)."""
import math

import torch

from .. import ops
from ..Riemannian_utils.sphere_utils_torch import rotate_along_geodesic


def _no_graph(points, sphere_axes, sphere_distances):
    """True when no autograd graph is wanted through the mapping (then the fused all-levels launches serve) and a HIP device is there"""
    if not (points.is_cuda or torch.cuda.is_available()):
        return False
    if not torch.is_grad_enabled():
        return True
    return not (points.requires_grad or any(torch.is_tensor(t) and t.requires_grad for t in list(sphere_axes) + list(sphere_distances)))


def _dist_value(sphere_distance_to_axis):
    return float(sphere_distance_to_axis.reshape(-1)[0]) if torch.is_tensor(sphere_distance_to_axis) else float(sphere_distance_to_axis)


def _north(like_axis):
    north = torch.zeros_like(like_axis.reshape(-1))
    north[-1] = 1.0
    return north


def _to_north(points, sphere_axis):
    """points rotated by the rotation that carries the axis to the north pole (nested_spheres_utils.py:33-36): a rank-2 update of the
    points - two inner products per point, no d x d matrix, no GEMM"""
    axis = sphere_axis.reshape(-1).to(points.device, points.dtype)
    return rotate_along_geodesic(points, axis, _north(axis))


def _from_north(points, sphere_axis):
    axis = sphere_axis.reshape(-1).to(points.device, points.dtype)
    return rotate_along_geodesic(points, axis, _north(axis), inverse=True)


def projection_from_sphere_to_nested_sphere(x, sphere_axis, sphere_distance_to_axis):
    """Points of S^d projected onto the small circle at distance `sphere_distance_to_axis` from `sphere_axis`
    (nested_spheres_utils.py:13-65).  x: (N, d) or (..., N, d)."""
    flat = x.reshape(-1, x.shape[-1])
    y_rot = ops.nested_sphere_epilogue(_to_north(flat, sphere_axis).detach(), _dist_value(sphere_distance_to_axis), mode=1).to(x.dtype)
    return _from_north(y_rot, sphere_axis).reshape(x.shape)


def projection_from_sphere_to_next_subsphere(x, sphere_axis, sphere_distance_to_axis):
    """S^d -> S^(d-1) (nested_spheres_utils.py:68-114); differentiable in x and in the axis."""
    flat = x.reshape(-1, x.shape[-1])
    z = ops.nested_sphere_next(_to_north(flat, sphere_axis), _dist_value(sphere_distance_to_axis))
    return z.reshape(tuple(x.shape[:-1]) + (x.shape[-1] - 1,))


def projection_from_sphere_to_subsphere(x, sphere_axes, sphere_distances_to_axes):
    """[x, x_{d-1}, ..., x_{d-r}] (nested_spheres_utils.py:117-146)."""
    if not isinstance(sphere_axes, list):
        sphere_axes = [sphere_axes]
    if not isinstance(sphere_distances_to_axes, list):
        sphere_distances_to_axes = [sphere_distances_to_axes]
    if _no_graph(x, sphere_axes, sphere_distances_to_axes) and x.dim() == 2 and len(sphere_axes) >= 1:
        # nothing to differentiate: every level for every point in ONE launch (gabo_nested_sphere_project)
        return [t.to(x.device, x.dtype) for t in ops.nested_sphere_project_all(x, sphere_axes, sphere_distances_to_axes)]
    x_subsphere = [x]
    for axis, dist in zip(sphere_axes, sphere_distances_to_axes):
        x_subsphere.append(projection_from_sphere_to_next_subsphere(x_subsphere[-1], axis, dist))
    return x_subsphere


def projection_from_subsphere_to_next_sphere(x_subsphere, sphere_axis, sphere_distance_to_axis):
    """S^(d-1) -> S^d: [sin r z, cos r] rotated from the north pole to the axis (nested_spheres_utils.py:149-179).  A concatenation
    and one rank-2 update: torch on the inputs' device."""
    if torch.is_tensor(sphere_distance_to_axis):          # kept as a tensor: the reconstruction cost is differentiated w.r.t. it
        r = sphere_distance_to_axis.reshape(()).to(x_subsphere.device, x_subsphere.dtype)
        sin_r, cos_r = torch.sin(r), torch.cos(r)
    else:
        sin_r, cos_r = math.sin(float(sphere_distance_to_axis)), math.cos(float(sphere_distance_to_axis))
    cos_vector = cos_r * torch.ones(x_subsphere.shape[0], 1, dtype=x_subsphere.dtype, device=x_subsphere.device)
    # R(north -> axis) = R(axis -> north)^T
    return _from_north(torch.cat((sin_r * x_subsphere, cos_vector), 1), sphere_axis)


def projection_from_subsphere_to_sphere(x_subsphere, sphere_axes, sphere_distances_to_axes):
    """[x_subsphere, x_{d-r+1}, ..., x_d], axes used in reverse order (nested_spheres_utils.py:182-218)."""
    if not isinstance(sphere_axes, list):
        sphere_axes = [sphere_axes]
    if not isinstance(sphere_distances_to_axes, list):
        sphere_distances_to_axes = [sphere_distances_to_axes]
    if _no_graph(x_subsphere, sphere_axes, sphere_distances_to_axes) and x_subsphere.dim() == 2 and len(sphere_axes) >= 1:
        return [t.to(x_subsphere.device, x_subsphere.dtype) for t in ops.nested_sphere_lift_all(x_subsphere, sphere_axes, sphere_distances_to_axes)]
    x = [x_subsphere]
    nb = len(sphere_axes)
    for s in range(nb):
        x.append(projection_from_subsphere_to_next_sphere(x[-1], sphere_axes[nb - s - 1], sphere_distances_to_axes[nb - s - 1]))
    return x
