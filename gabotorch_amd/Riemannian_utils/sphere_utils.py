"""numpy-facing sphere helpers with the reference's names and argument order (BoManifolds/Riemannian_utils/sphere_utils.py:14-90),
computed by the batched HIP sphere-manifold kernel.  Points are (dim,) vectors or (dim, N) column stacks, as in the reference."""
import numpy as np
import torch

from .. import _lib, ops


def _rows(a):
    a = np.asarray(a, dtype=float)
    return (a[None], True) if a.ndim < 2 else (a.T, False)


def expmap(u, x0):
    """Exp_{x0}(u) -> (dim, N)"""
    U, _ = _rows(u)
    X0, _ = _rows(x0)
    return ops.sphere_manifold_op(_lib.GABO_SPH_EXP, torch.as_tensor(X0), torch.as_tensor(U)).numpy().T


def logmap(x, x0):
    """Log_{x0}(x) -> (dim, N)"""
    X, _ = _rows(x)
    X0, _ = _rows(x0)
    return ops.sphere_manifold_op(_lib.GABO_SPH_LOG, torch.as_tensor(X0), torch.as_tensor(X)).numpy().T


def sphere_distance(x, y):
    X, _ = _rows(x)
    Y, _ = _rows(y)
    return ops.sphere_manifold_op(_lib.GABO_SPH_DIST, torch.as_tensor(X), torch.as_tensor(Y)).numpy()


def rotation_from_sphere_points(x, y):
    """The rotation matrix that carries the unit vector x to the unit vector y along their geodesic (sphere_utils.py:168-202; used by
    `examples/bo_sphere/constrained_benchmark_examples/gabo_sphere_inequality_constraints.py`).  It acts in span{x, y} only: with t = <x, y>
    (clipped to [-1, 1]), s = sin(acos t) and e the unit vector of that plane orthogonal to y,
        R = I + s (y e^T - e y^T) + (t - 1) (y y^T + e e^T)
    (Jung, Dryden & Marron 2012, appendix).  Host-side numpy: one d x d matrix per call, not on the device path (the batched, matrix-free
    form the nested-sphere kernels use is `sphere_utils_torch.rotate_along_geodesic`)."""
    x = np.asarray(x, dtype=float).reshape(-1)
    y = np.asarray(y, dtype=float).reshape(-1)
    t = float(np.clip(x @ y, -1.0, 1.0))
    e = x - t * y
    e = e / np.linalg.norm(e)
    s = np.sqrt((1.0 - t) * (1.0 + t))
    ye = np.outer(y, e)
    return np.eye(x.size) + s * (ye - ye.T) + (t - 1.0) * (np.outer(y, y) + np.outer(e, e))
