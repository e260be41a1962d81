"""Torch-facing sphere utilities with the reference's names (BoManifolds/Riemannian_utils/sphere_utils_torch.py)."""
from .. import _lib, ops


def sphere_distance_torch(x1, x2, diag=False):
    """acos(clamp(<x1_i, x2_j>))   (sphere_utils_torch.py:12-55)."""
    return ops.sphere_kernel(x1, x2, 1.0, _lib.GABO_OUT_DISTANCE, diag=diag)


def rotation_from_sphere_points_torch(x, y):
    """Rotation matrix moving x to y along the geodesic (sphere_utils_torch.py:58-93; Jung et al. 2012, appendix).  A d x d
    matrix built from two vectors: plain torch on the inputs' device, differentiable in x and y."""
    import torch
    if x.dim() == 1:
        x = x.unsqueeze(-2)
    if y.dim() == 1:
        y = y.unsqueeze(-2)
    dim = x.shape[1]
    inner = torch.mm(x, y.T).clamp(-1.0 + 1e-15, 1.0 - 1e-15)
    c_vec = x - y * inner
    c_vec = c_vec / torch.norm(c_vec)
    eye = torch.eye(dim, dtype=inner.dtype, device=inner.device)
    return eye + torch.sin(torch.acos(inner)) * (torch.mm(y.T, c_vec) - torch.mm(c_vec.T, y)) \
        + (inner - 1.0) * (torch.mm(y.T, y) + torch.mm(c_vec.T, c_vec))
