"""Torch-facing sphere utilities with the reference's names (BoManifolds/Riemannian_utils/sphere_utils_torch.py)."""
from .. import _lib, ops


def sphere_distance_torch(x1, x2, diag=False):
    """acos(clamp(<x1_i, x2_j>))   (sphere_utils_torch.py:12-55)."""
    return ops.sphere_kernel(x1, x2, 1.0, _lib.GABO_OUT_DISTANCE, diag=diag)
