"""Nested SPD projection with the reference's name and signature (BoManifolds/nested_mappings/nested_spd_utils.py:13-48)."""
from .. import ops
from ..Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch


def projection_from_spd_to_nested_spd(x_spd, projection_matrix):
    """Y = W^T X W for X (..., D, D) -> (..., d, d), computed by gabo_spd_project (Mandel in / Mandel out)."""
    y = ops.spd_project(symmetric_matrix_to_vector_mandel_torch(x_spd.detach()), projection_matrix.detach())
    return vector_to_symmetric_matrix_mandel_torch(y)
