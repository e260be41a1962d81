"""Objectives of HD-GaBO on spheres (`BoManifolds/BO_test_functions/nested_test_functions_sphere.py:13-70`): a test function of the subsphere S^m
evaluated at the nested-sphere projection of a point of S^d."""
import torch

from ..nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere, projection_from_subsphere_to_sphere


def nested_function_sphere(x, subsphere_manifold, test_function, sphere_axes, sphere_distances_to_axes):
    if x.dim() < 2:
        x = x[None]
    return test_function(projection_from_sphere_to_subsphere(x, sphere_axes, sphere_distances_to_axes)[-1], subsphere_manifold)


def optimum_nested_function_sphere(optimum_function, subsphere_manifold, sphere_axes, sphere_distances_to_axes):
    """The minimiser is unique on the subsphere only; the point of S^d returned is the one lying on the nested sphere itself (:39-70)."""
    nested_x, y = optimum_function(subsphere_manifold)
    nested_x = torch.as_tensor(nested_x, dtype=sphere_axes[0].dtype, device=sphere_axes[0].device)
    x = projection_from_subsphere_to_sphere(nested_x, sphere_axes, sphere_distances_to_axes)[-1]
    return x.detach().cpu().numpy(), y
