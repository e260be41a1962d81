"""Manifold objects with pymanopt's duck type (the reference drives its solvers through exactly these methods:
SURVEY 8a/a9), backed by the batched HIP kernels.  Every method accepts a single point (d, d) / (dim,) or a batch
(R, d, d) / (R, dim) of restarts, as numpy arrays or torch tensors, and returns the same kind of object it was given.

`rand` stays a host callable on purpose: callers monkey-patch it (examples/gabo_spd.py:102 assigns `spd_sample`).
"""
import numpy as np
import torch

from . import _lib, ops


def _wrap(fn):
    def inner(*args):
        np_in = any(isinstance(a, np.ndarray) for a in args)
        targs = [torch.as_tensor(a) if isinstance(a, np.ndarray) else a for a in args]
        out = fn(*targs)
        if np_in:
            res = tuple(o.cpu().numpy() for o in out) if isinstance(out, tuple) else out.cpu().numpy()
            ops.check_deferred()          # (numpy in, numpy out: the copy above waited for the device - a launch that failed is reported here)
            return res
        return out
    return inner


def _haar_q(n, k, gen):
    """k Haar-distributed orthogonal n x n matrices (up to column signs), laid out n x n x k (batch LAST: every numpy operation below runs
    over a contiguous axis of length k).  The Q of qr(randn(n, n)) - what the reference's sampler uses (spd_utils.py:290-306) - is the product of
    n - 1 Householder reflections whose vectors are, by the rotation invariance of the Gaussian, independent Gaussian vectors of lengths
    n, n - 1, ..., 2 (the subgroup algorithm of Diaconis and Shahshahani); drawing those vectors directly gives the same distribution without
    factoring anything: n (n + 1) / 2 - 1 normals per matrix instead of n^2, and no LAPACK call per matrix (numpy's batched qr takes 2-3 ms for
    2048 5 x 5 matrices - the largest host item of a sweep with host-drawn raw samples).  Q diag(lam) Q^T does not see the column signs."""
    q = np.zeros((n, n, k))
    q[np.arange(n), np.arange(n)] = 1.0
    for j in range(n - 1):
        v = gen.standard_normal((n - j, k))
        nrm = np.sqrt((v * v).sum(0))
        v[0] += np.where(v[0] >= 0, nrm, -nrm)
        vn = (v * v).sum(0)
        v *= np.sqrt(2.0) / np.sqrt(np.where(vn > 0, vn, 1.0))             # H = I - v v^T with |v|^2 = 2
        q[:, j:] -= (q[:, j:] * v[None]).sum(1)[:, None] * v[None]
    return q


class PositiveDefinite:
    """S^n_++ with the affine-invariant metric ([3P] pymanopt.manifolds.PositiveDefinite)."""

    batched = True        # every method takes one point or a batch of restarts

    def __init__(self, n):
        self._n = n
        self._shape = (n, n)
        self.min_eig, self.max_eig = 1.0, 2.0

    @property
    def dim(self):
        return self._n * (self._n + 1) // 2

    @property
    def typicaldist(self):
        return float(np.sqrt(self.dim))

    def rand(self):
        """[3P] eigenvalues U[1,2], orthogonal factor from qr(randn); numpy global RNG.  Usually replaced by spd_sample."""
        lam = 1.0 + np.random.rand(self._n)
        q, _ = np.linalg.qr(np.random.randn(self._n, self._n))
        return q @ np.diag(lam) @ q.T

    def zerovec(self, x):
        return np.zeros_like(x) if isinstance(x, np.ndarray) else torch.zeros_like(x)

    def rand_batch(self, k):
        """k samples of the `spd_sample` distribution (eigenvalues U[min_eig, max_eig], Haar-like Q from qr(randn)) in one
        vectorised numpy call.  Same distribution as k calls of spd_utils.spd_sample, NOT the same draw order from the global
        RNG - used only when the caller opts in (options={"batched_rand": True})."""
        lam = self.min_eig + (self.max_eig - self.min_eig) * np.random.rand(k, self._n)
        # the normals from a PCG64 / ziggurat stream seeded from the GLOBAL stream (np.random.seed still fixes the draw): 0.25 ms for
        # 2048 5 x 5 matrices where the global stream's polar method takes 0.75
        gen = np.random.Generator(np.random.PCG64(int(np.random.randint(0, 2 ** 31 - 1))))
        q = _haar_q(self._n, k, gen)                                                  # n x n x k: batch last
        m = np.einsum("ick,jck->ijk", q * lam.T[None], q)                             # Q diag(lam) Q^T
        m = np.ascontiguousarray(m.transpose(2, 0, 1))
        return 0.5 * (m + m.transpose(0, 2, 1))

    def rand_batch_device(self, k, device, first=0, count=None):
        """k samples of the same distribution drawn on the device (gabo_spd_sample); the seed comes from numpy's global RNG, so
        np.random.seed(...) makes the draw reproducible.  Used when the caller opts in (options={"device_rand": True}).
        `first` / `count`: only samples first ... first + count - 1 of the k are drawn (the stream is addressed by sample index, so the
        shards of ranks that share the numpy seed add up to exactly the k samples one rank would draw: SURVEY 8e)."""
        seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
        return ops.spd_sample(k - first if count is None else count, self._n, self.min_eig, self.max_eig, seed, device, first=first)

    exp = staticmethod(_wrap(lambda x, u: ops.spd_manifold_op(_lib.GABO_SPD_EXP, x, u)))
    retr = exp                                                                       # [3P] retr = exp
    log = staticmethod(_wrap(lambda x, y: ops.spd_manifold_op(_lib.GABO_SPD_LOG, x, y)))
    inner = staticmethod(_wrap(lambda x, u, v: ops.spd_manifold_op(_lib.GABO_SPD_INNER, x, u, v)))
    norm = staticmethod(_wrap(lambda x, u: ops.spd_manifold_op(_lib.GABO_SPD_NORM, x, u)))
    dist = staticmethod(_wrap(lambda x, y: ops.spd_manifold_op(_lib.GABO_SPD_DIST, x, y)))
    egrad2rgrad = staticmethod(_wrap(lambda x, g: ops.spd_manifold_op(_lib.GABO_SPD_EGRAD2RGRAD, x, g)))
    ehess2rhess = staticmethod(_wrap(lambda x, eg, eh, u: ops.spd_manifold_op(_lib.GABO_SPD_EHESS2RHESS, x, eg, eh, u)))

    @staticmethod
    def proj(x, u):
        return 0.5 * (u + (u.swapaxes(-1, -2) if isinstance(u, np.ndarray) else u.transpose(-1, -2)))

    @staticmethod
    def transp(x1, x2, d):
        return d                                                                      # [3P] identity transport


class Sphere:
    """S^{n-1} in R^n ([3P] pymanopt.manifolds.Sphere)."""

    batched = True

    def __init__(self, n):
        self._n = n
        self._shape = (n,)

    @property
    def dim(self):
        return self._n - 1

    @property
    def typicaldist(self):
        return float(np.pi)

    def rand(self):
        x = np.random.randn(self._n)
        return x / np.linalg.norm(x)

    def rand_batch(self, k):
        # (the normals from a PCG64 / ziggurat stream seeded from the GLOBAL stream, as PositiveDefinite.rand_batch: np.random.seed still fixes the draw;
        # 2048 x 10 deviates in 0.06 ms where the global stream's polar method takes 0.2)
        gen = np.random.Generator(np.random.PCG64(int(np.random.randint(0, 2 ** 31 - 1))))
        x = gen.standard_normal((k, self._n))
        return x / np.sqrt(np.einsum("ij,ij->i", x, x))[:, None]

    def zerovec(self, x):
        return np.zeros_like(x) if isinstance(x, np.ndarray) else torch.zeros_like(x)

    @staticmethod
    def inner(x, u, v):
        return (u * v).sum(-1)

    @staticmethod
    def norm(x, u):
        return np.sqrt((u * u).sum(-1)) if isinstance(u, np.ndarray) else (u * u).sum(-1).sqrt()

    proj = staticmethod(_wrap(lambda x, h: ops.sphere_manifold_op(_lib.GABO_SPH_PROJ, x, h)))
    egrad2rgrad = proj
    retr = staticmethod(_wrap(lambda x, u: ops.sphere_manifold_op(_lib.GABO_SPH_RETR, x, u)))
    exp = staticmethod(_wrap(lambda x, u: ops.sphere_manifold_op(_lib.GABO_SPH_EXP, x, u)))
    log = staticmethod(_wrap(lambda x, y: ops.sphere_manifold_op(_lib.GABO_SPH_LOG, x, y)))
    dist = staticmethod(_wrap(lambda x, y: ops.sphere_manifold_op(_lib.GABO_SPH_DIST, x, y)))
    ehess2rhess = staticmethod(_wrap(lambda x, eg, eh, u: ops.sphere_manifold_op(_lib.GABO_SPH_EHESS2RHESS, x, eg, eh, u)))
    transp = staticmethod(_wrap(lambda x, y, u: ops.sphere_manifold_op(_lib.GABO_SPH_PROJ, y, u)))
