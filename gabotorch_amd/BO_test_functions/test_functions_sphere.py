"""Objectives on the sphere under the reference's module path (BoManifolds/BO_test_functions/test_functions_sphere.py)."""
import numpy as np
import torch

from .test_functions import ackley_function_sphere  # noqa: F401


def optimum_ackley_sphere(sphere_manifold):
    """(location, value) of the global minimum: the base point (1, 0, ..., 0) (test_functions_sphere.py:68-91)"""
    opt_x = np.zeros((1, sphere_manifold._shape[0]))
    opt_x[0, 0] = 1.0
    return opt_x, ackley_function_sphere(torch.tensor(opt_x), sphere_manifold).numpy()
