"""Nested SPD projection with the reference's name and signature (BoManifolds/nested_mappings/nested_spd_utils.py:13-48)."""
from .. import ops
from ..Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch


def projection_from_spd_to_nested_spd(x_spd, projection_matrix):
    """Y = W^T X W for X (..., D, D) -> (..., d, d), computed by gabo_spd_project (Mandel in / Mandel out)."""
    y = ops.spd_project_diff(symmetric_matrix_to_vector_mandel_torch(x_spd), projection_matrix)
    return vector_to_symmetric_matrix_mandel_torch(y)


def projection_from_nested_spd_to_spd(x_spd_low_dimension, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                      contraction_matrix, sqrt_low=None):
    """Approximate right inverse of `projection_from_spd_to_nested_spd` (nested_spd_utils.py:51-118): with R = [W, V],
    Xr = [[Y, B], [B^T, C]], B = Y^1/2 K C^1/2, X = R Xr R^T.  Y: (d, d) or (N, d, d).  The matrix square roots are one
    batched HIP launch (GABO_SPD_SQRTM, differentiable through gabo_spd_matfun_backward); the block assembly and the two small
    products are torch on the inputs' device.  sqrt_low (extension): Y^1/2 when the caller already has it (the reconstruction
    optimiser evaluates this map thousands of times for fixed Y)."""
    import torch

    from .. import _lib
    y = x_spd_low_dimension
    single = y.dim() == 2
    if single:
        y = y.unsqueeze(0)
    dev, dt = y.device, y.dtype
    W, V = projection_matrix.to(dev, dt), projection_complement_matrix.to(dev, dt)
    C, K = bottom_spd_matrix.to(dev, dt), contraction_matrix.to(dev, dt)
    R = torch.cat((W, V), dim=1)
    if C.requires_grad:
        sqrt_c = ops.spd_matrix_function(C, _lib.GABO_SPD_SQRTM).to(dt)    # differentiable (reconstruction costs, f4)
    else:
        # a fixed bottom matrix (the latent constraints evaluate this map thousands of times per sweep): its square root - an
        # eigen-solve of a (D-d) x (D-d) matrix - is computed once and kept on the tensor
        memo = getattr(bottom_spd_matrix, "_gabo_sqrt", None)
        if memo is None or memo[0] != bottom_spd_matrix._version or memo[1].device != dev or memo[1].dtype != dt:
            memo = (bottom_spd_matrix._version, ops.spd_matrix_function(C, _lib.GABO_SPD_SQRTM).to(dt))
            try:
                bottom_spd_matrix._gabo_sqrt = memo
            except AttributeError:      # (not every tensor subclass takes attributes)
                pass
        sqrt_c = memo[1]
    if sqrt_low is None:
        sqrt_y = ops.spd_matrix_function(y, _lib.GABO_SPD_SQRTM).to(dt)
    else:
        sqrt_y = (sqrt_low.unsqueeze(0) if sqrt_low.dim() == 2 else sqrt_low).to(dev, dt)
    side = sqrt_y @ K @ sqrt_c
    n = y.shape[0]
    top = torch.cat((y, side), dim=2)
    bottom = torch.cat((side.transpose(1, 2), C.expand(n, *C.shape)), dim=2)
    xr = torch.cat((top, bottom), dim=1)
    x = R @ xr @ R.T
    return x[0] if single else x
