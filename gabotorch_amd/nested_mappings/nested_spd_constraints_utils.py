"""Constraints of the latent acquisition optimisation of HD-GaBO on S^D_++, with the reference's names and signatures
(BoManifolds/nested_mappings/nested_spd_constraints_utils.py:14-100): the eigenvalue bounds are stated in the ORIGINAL space, so the
nested point is lifted with projection_from_nested_spd_to_spd and the extreme eigenvalue of the result is bounded.  Single (d, d)
points or batches (R, d, d); differentiable (sqrtm adjoint + v v^T eigenvalue gradient, all HIP)."""
import torch

from ..Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch, min_eigenvalue_constraint_torch
from .nested_spd_utils import projection_from_nested_spd_to_spd, projection_from_spd_to_nested_spd


def max_eigenvalue_nested_spd_constraint(x_nested_spd, maximum_eigenvalue, projection_matrix, projection_complement_matrix,
                                         bottom_spd_matrix, contraction_matrix):
    """maximum_eigenvalue - lambda_max(reconstruction(x_nested_spd))   (nested_spd_constraints_utils.py:14-42)."""
    x_spd = projection_from_nested_spd_to_spd(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    return max_eigenvalue_constraint_torch(x_spd, maximum_eigenvalue)


def min_eigenvalue_nested_spd_constraint(x_nested_spd, minimum_eigenvalue, projection_matrix, projection_complement_matrix,
                                         bottom_spd_matrix, contraction_matrix):
    """lambda_min(reconstruction(x_nested_spd)) - minimum_eigenvalue   (nested_spd_constraints_utils.py:45-73)."""
    x_spd = projection_from_nested_spd_to_spd(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    return min_eigenvalue_constraint_torch(x_spd, minimum_eigenvalue)


def random_nested_spd_with_spd_eigenvalue_constraints(self, random_spd_fct, projection_matrix):
    """A nested-SPD sample = the projection of a sample of the original space (nested_spd_constraints_utils.py:76-100); bound to
    the latent manifold as its `rand` (functools.partial + types.MethodType, examples/hd_gabo_spd.py:239-242): numpy out."""
    x_spd = torch.as_tensor(random_spd_fct(), dtype=projection_matrix.dtype)
    return projection_from_spd_to_nested_spd(x_spd.to(projection_matrix.device), projection_matrix).cpu().numpy()
