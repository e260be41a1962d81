"""numpy-facing SPD helpers with the reference's names and argument order (BoManifolds/Riemannian_utils/spd_utils.py).
The Mandel maps and `spd_sample` are host numpy (they feed host callbacks and a host RNG: SURVEY 8b); the exp/log maps and
the distance go through the HIP manifold kernels."""
import numpy as np
import torch

from .. import _lib, ops


def _mandel_index(d):
    r, c = [], []
    for k in range(d):
        for i in range(d - k):
            r.append(i)
            c.append(i + k)
    return np.array(r), np.array(c)


def symmetric_matrix_to_vector_mandel(M):
    """(spd_utils.py:57-76)"""
    M = np.asarray(M)
    r, c = _mandel_index(M.shape[-1])
    return M[..., r, c] * np.where(r == c, 1.0, 2.0 ** 0.5)


def vector_to_symmetric_matrix_mandel(v):
    """(spd_utils.py:79-101)"""
    v = np.asarray(v, dtype=float)
    d = int((-1.0 + (1.0 + 8.0 * v.shape[-1]) ** 0.5) / 2.0)
    r, c = _mandel_index(d)
    s = np.where(r == c, 1.0, 1.0 / 2.0 ** 0.5)
    M = np.zeros(v.shape[:-1] + (d, d))
    M[..., r, c] = v * s
    M[..., c, r] = v * s
    return M


def expmap(U, S):
    """Exp_S(U)   (spd_utils.py:104-120; tangent first, base second)"""
    return ops.spd_manifold_op(_lib.GABO_SPD_EXP, torch.as_tensor(np.asarray(S)), torch.as_tensor(np.asarray(U))).numpy()


def logmap(X, S):
    """Log_S(X)   (spd_utils.py:123-139; point first, base second)"""
    return ops.spd_manifold_op(_lib.GABO_SPD_LOG, torch.as_tensor(np.asarray(S)), torch.as_tensor(np.asarray(X))).numpy()


def affine_invariant_distance(S1, S2):
    """(spd_utils.py:180-197)"""
    return float(ops.spd_manifold_op(_lib.GABO_SPD_DIST, torch.as_tensor(np.asarray(S1)), torch.as_tensor(np.asarray(S2))))


def spd_sample(self):
    """Random SPD matrix with eigenvalues U[self.min_eig, self.max_eig]; numpy GLOBAL RNG, the reference's draw order
    (spd_utils.py:290-306).  Meant to be bound as `manifold.rand` (examples/gabo_spd.py:102)."""
    d = self.min_eig * np.ones(1) + (self.max_eig - self.min_eig) * np.random.rand(self._n)
    u, _ = np.linalg.qr(np.random.randn(self._n, self._n))
    return np.dot(u, np.dot(np.diag(d), u.T))
