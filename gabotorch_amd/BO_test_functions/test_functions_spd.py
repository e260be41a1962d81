"""Objectives on the SPD manifold under the reference's module path (BoManifolds/BO_test_functions/test_functions_spd.py)."""
import numpy as np
import torch

from ..Riemannian_utils.spd_utils import symmetric_matrix_to_vector_mandel
from .test_functions import ackley_function_spd, get_ackley_base, get_rosenbrock_base, rosenbrock_function_spd  # noqa: F401


def optimum_ackley_spd(spd_manifold):
    """(location, value) of the global minimum: the base point (test_functions_spd.py:72-92)"""
    opt_x = get_ackley_base(spd_manifold)
    return opt_x, ackley_function_spd(torch.tensor(symmetric_matrix_to_vector_mandel(opt_x)[None]), spd_manifold).numpy()


def optimum_rosenbrock_spd(spd_manifold):
    """(location, value) of the global minimum: tangent coordinates (1, ..., 1) at the base point, mapped back with the manifold
    exponential (test_functions_spd.py:165-190)"""
    d = spd_manifold._n
    base = get_rosenbrock_base(spd_manifold)
    from ..Riemannian_utils.spd_utils import vector_to_symmetric_matrix_mandel
    voigt = np.ones(d * (d + 1) // 2)
    voigt[d:] *= 2.0 ** 0.5                                      # Voigt -> Mandel
    opt_x = np.asarray(spd_manifold.exp(base, vector_to_symmetric_matrix_mandel(voigt)))
    return opt_x, rosenbrock_function_spd(torch.tensor(symmetric_matrix_to_vector_mandel(opt_x)[None]), spd_manifold).numpy()
