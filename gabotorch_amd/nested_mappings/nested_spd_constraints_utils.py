"""Constraints of the latent acquisition optimisation of HD-GaBO on S^D_++, with the reference's names and signatures
(BoManifolds/nested_mappings/nested_spd_constraints_utils.py:14-100): the eigenvalue bounds are stated in the ORIGINAL space, so the
nested point is lifted with projection_from_nested_spd_to_spd and the extreme eigenvalue of the result is bounded.  Single (d, d)
points or batches (R, d, d); differentiable (sqrtm adjoint + v v^T eigenvalue gradient, all HIP)."""
import torch

from .. import ops
from ..Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch, min_eigenvalue_constraint_torch
from .nested_spd_utils import projection_from_nested_spd_to_spd, projection_from_spd_to_nested_spd


def _lifted_extremes(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix, contraction_matrix):
    """(lambda_max, lambda_min) of the lifted point(s) by ONE launch (gabo_nested_spd_extreme_eigenvalues), or None when the mapping
    itself is being differentiated (then the composed, fully differentiable path below serves).  What the launch needs of the mapping -
    X0 = V C V^T and P = V (K C^1/2)^T - is computed once per (W, V, C, K) and kept on the bottom matrix tensor."""
    params = (projection_matrix, projection_complement_matrix, bottom_spd_matrix, contraction_matrix)
    if not all(torch.is_tensor(t) for t in params) or any(t.requires_grad for t in params) or not torch.is_tensor(x_nested_spd):
        return None
    if not (x_nested_spd.is_cuda or any(t.is_cuda for t in params) or torch.cuda.is_available()):
        return None
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in params)
    memo = getattr(bottom_spd_matrix, "_gabo_lift", None)
    if memo is None or memo[0] != key:
        memo = (key, ops.nested_spd_lift_prepare(*params))
        try:
            bottom_spd_matrix._gabo_lift = memo
        except AttributeError:          # (not every tensor subclass takes attributes)
            pass
    w, x0, p = memo[1]
    return ops.nested_spd_extremes(x_nested_spd, w, p, x0)


def max_eigenvalue_nested_spd_constraint(x_nested_spd, maximum_eigenvalue, projection_matrix, projection_complement_matrix,
                                         bottom_spd_matrix, contraction_matrix):
    """maximum_eigenvalue - lambda_max(reconstruction(x_nested_spd))   (nested_spd_constraints_utils.py:14-42)."""
    lam = _lifted_extremes(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix, contraction_matrix)
    if lam is not None:
        return maximum_eigenvalue - lam[..., 0]
    x_spd = projection_from_nested_spd_to_spd(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    return max_eigenvalue_constraint_torch(x_spd, maximum_eigenvalue)


def min_eigenvalue_nested_spd_constraint(x_nested_spd, minimum_eigenvalue, projection_matrix, projection_complement_matrix,
                                         bottom_spd_matrix, contraction_matrix):
    """lambda_min(reconstruction(x_nested_spd)) - minimum_eigenvalue   (nested_spd_constraints_utils.py:45-73)."""
    lam = _lifted_extremes(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix, contraction_matrix)
    if lam is not None:
        return lam[..., 1] - minimum_eigenvalue
    x_spd = projection_from_nested_spd_to_spd(x_nested_spd, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    return min_eigenvalue_constraint_torch(x_spd, minimum_eigenvalue)


def random_nested_spd_with_spd_eigenvalue_constraints(self, random_spd_fct, projection_matrix):
    """A nested-SPD sample = the projection of a sample of the original space (nested_spd_constraints_utils.py:76-100); bound to
    the latent manifold as its `rand` (functools.partial + types.MethodType, examples/hd_gabo_spd.py:239-242): numpy in (the host
    sampler's matrix), numpy out.  One D x D sample is projected where it lives - W^T X W on the host copy of W, kept on the tensor -
    instead of a host -> device -> host round trip per raw sample (1024 of them per sweep); batches of device-resident points go through
    projection_from_spd_to_nested_spd / gabo_spd_project."""
    import numpy as np
    memo = getattr(projection_matrix, "_gabo_host", None)
    if memo is None or memo[0] != projection_matrix._version:
        memo = (projection_matrix._version, projection_matrix.detach().cpu().numpy().astype(np.float64))
        try:
            projection_matrix._gabo_host = memo
        except AttributeError:
            pass
    w = memo[1]
    x_spd = np.asarray(random_spd_fct(), dtype=np.float64)
    y = w.T @ x_spd @ w
    return 0.5 * (y + y.T)
