"""Benchmark objectives for the end-to-end configurations (BASELINE.json configs 1, 4, 5): Ackley and Rosenbrock defined in the
tangent space of a base point and pulled back through the manifold logarithm, as in the reference
(BoManifolds/BO_test_functions/test_functions_spd.py:14-162, test_functions_sphere.py:12-65).  Host numpy on purpose: an
objective evaluation is one point per BO iteration.  `manifold.log(base, x)` is the only manifold call (pymanopt argument order).
"""
import numpy as np
import torch

from ..Riemannian_utils.spd_utils import symmetric_matrix_to_vector_mandel, vector_to_symmetric_matrix_mandel


def _ackley(v):
    a, b, c = 20, 0.2, 2 * np.pi
    n = v.shape[0]
    return -a * np.exp(-b * np.sqrt(np.sum(v ** 2) / n)) - np.exp(np.sum(np.cos(c * v) / n)) + a + np.exp(1.0)


def _tangent_coordinates_spd(x, spd_manifold, base):
    d = spd_manifold._n
    x = x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    if np.ndim(x) < 2:
        x = x[None]
    xm = vector_to_symmetric_matrix_mandel(x[0])
    v = symmetric_matrix_to_vector_mandel(np.asarray(spd_manifold.log(base, xm)))
    v[d:] /= 2.0 ** 0.5          # Voigt instead of Mandel: every distinct element once (test_functions_spd.py:57-58)
    return v


def get_ackley_base(spd_manifold):
    return 2 * np.eye(spd_manifold._n)


def get_rosenbrock_base(spd_manifold):
    return 2 * np.eye(spd_manifold._n)


def ackley_function_spd(x, spd_manifold):
    """-> torch [1,1]   (test_functions_spd.py:14-69)"""
    y = _ackley(_tangent_coordinates_spd(x, spd_manifold, get_ackley_base(spd_manifold)))
    return torch.tensor(np.array(y)[None, None], dtype=x.dtype if torch.is_tensor(x) else torch.float64)


def rosenbrock_function_spd(x, spd_manifold):
    """-> torch [1,1]   (test_functions_spd.py:113-162)"""
    v = _tangent_coordinates_spd(x, spd_manifold, get_rosenbrock_base(spd_manifold))
    y = np.sum(100 * (v[1:] - v[:-1] ** 2) ** 2 + (1 - v[:-1]) ** 2)
    return torch.tensor(np.array(y)[None, None], dtype=x.dtype if torch.is_tensor(x) else torch.float64)


def ackley_function_sphere(x, sphere_manifold):
    """Base point (1, 0, ..., 0); the first tangent coordinate is dropped   (test_functions_sphere.py:12-65)."""
    dt = x.dtype if torch.is_tensor(x) else torch.float64
    x = x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    if np.ndim(x) < 2:
        x = x[None]
    n = sphere_manifold._shape[0]
    base = np.zeros((1, n))
    base[0, 0] = 1.0
    v = np.asarray(sphere_manifold.log(base, x))[0][1:]
    return torch.tensor(np.array(_ackley(v))[None, None], dtype=dt)
