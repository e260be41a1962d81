"""Objectives of HD-GaBO on SPD manifolds (`BoManifolds/BO_test_functions/nested_test_functions_spd.py:17-70`): a test function of the latent
manifold S^d_++ evaluated at the nested projection W^T X W of a point of S^D_++ (BASELINE config 5: Rosenbrock on S^2_++ inside S^20_++)."""
from ..nested_mappings.nested_spd_utils import projection_from_spd_to_nested_spd
from ..Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch


def projected_function_spd(x, low_dimensional_spd_manifold, test_function, projection_matrix):
    """x: Mandel vector(s) of S^D_++; projection_matrix: D x d with orthonormal columns.  Returns test_function(Mandel(W^T X W), latent manifold)."""
    latent = projection_from_spd_to_nested_spd(vector_to_symmetric_matrix_mandel_torch(x), projection_matrix)
    return test_function(symmetric_matrix_to_vector_mandel_torch(latent), low_dimensional_spd_manifold)


def optimum_projected_function_spd(optimum_function, low_dimensional_spd_manifold, projection_matrix):
    """(x*, f(x*)) on the LATENT manifold: the projection has no inverse, so the reference reports the latent minimiser (:49-70)."""
    return optimum_function(low_dimensional_spd_manifold)
