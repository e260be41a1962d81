"""Torch-facing SPD utilities with the reference's function names (BoManifolds/Riemannian_utils/spd_utils_torch.py),
computed by the HIP library."""
import torch

from .. import _lib, ops


def vector_to_symmetric_matrix_mandel_torch(vectors):
    """(..., d_vec) -> (..., d, d)   (spd_utils_torch.py:159-194)."""
    return _MandelToMatrix.apply(vectors)


def symmetric_matrix_to_vector_mandel_torch(matrices):
    """(..., d, d) -> (..., d_vec), both triangles averaged   (spd_utils_torch.py:197-226)."""
    return _MatrixToMandel.apply(matrices)


class _MandelToMatrix(torch.autograd.Function):
    # The two maps are linear and each is the other's adjoint:
    #   M_rc = M_cr = v_e / sqrt2  =>  dL/dv_e = (g_rc + g_cr) / sqrt2 = matrix_to_mandel(g)_e ;  diagonal: g_rr.
    @staticmethod
    def forward(ctx, v):
        return ops.mandel_to_matrix(v).to(v.dtype)

    @staticmethod
    def backward(ctx, g):
        return _MatrixToMandel.apply(g)


class _MatrixToMandel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m):
        return ops.matrix_to_mandel(m).to(m.dtype)

    @staticmethod
    def backward(ctx, g):
        # out_e = sqrt2/2 (m_rc + m_cr)  ->  d/dm_rc = d/dm_cr = g_e * sqrt2 / 2 = mandel_to_matrix(g)_rc
        return _MandelToMatrix.apply(g)


def affine_invariant_distance_torch(x1, x2, diagonal_distance=False):
    """x1 (..., N1, d, d), x2 (..., N2, d, d) SPD matrices -> (..., N1, N2)   (spd_utils_torch.py:53-121)."""
    if diagonal_distance is True:
        return torch.zeros(tuple(x2.shape[:-2]) + (1,), dtype=x1.dtype, device=x1.device)
    v1 = symmetric_matrix_to_vector_mandel_torch(x1)
    v2 = symmetric_matrix_to_vector_mandel_torch(x2)
    return ops.spd_ai_kernel(v1, v2, 1.0, _lib.GABO_OUT_DISTANCE)


def logm_torch(x):
    """Matrix logarithm of SPD matrices (..., d, d)   (spd_utils_torch.py:13-30); differentiable (first order)."""
    return ops.spd_matrix_function(x, _lib.GABO_SPD_LOGM)


def sqrtm_torch(x):
    """Matrix square root of SPD matrices (..., d, d)   (spd_utils_torch.py:33-50); differentiable (first order)."""
    return ops.spd_matrix_function(x, _lib.GABO_SPD_SQRTM)


def frobenius_distance_torch(x1, x2, diagonal_distance=False):
    """||x1_i - x2_j + 1e-15||_F for symmetric matrices x1 (..., N1, d, d), x2 (..., N2, d, d) -> (..., N1, N2)
    (spd_utils_torch.py:124-156); differentiable (first order) through the HIP backward."""
    if diagonal_distance is True:
        return torch.zeros(tuple(x2.shape[:-2]) + (1,), dtype=x1.dtype, device=x1.device)
    v1 = symmetric_matrix_to_vector_mandel_torch(x1)
    v2 = symmetric_matrix_to_vector_mandel_torch(x2)
    return ops.frobenius_kernel(v1, v2, 1.0, _lib.GABO_OUT_DISTANCE)
