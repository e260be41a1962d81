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
