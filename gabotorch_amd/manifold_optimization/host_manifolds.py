"""Host-side (numpy) manifolds for the GP HYPER-PARAMETERS: the handful of numbers the surrogate fit moves on a product of
Euclidean spaces, spheres (nested-sphere axes) and a Grassmannian (nested-SPD projection matrix) - the objects the reference
takes from pymanopt in manifold_gp_fit.py:163-173 ([3P] pymanopt 0.2.x semantics restated from memory).
They never see the data: the marginal likelihood and its gradient, where the time goes, run through the HIP kernels."""
import numpy as np


class Euclidean:
    def __init__(self, *shape):
        self._shape = tuple(shape)
        self.dim = int(np.prod(shape))
        self.typicaldist = float(np.sqrt(self.dim))

    def inner(self, x, u, v):
        return float(np.vdot(u, v))

    def norm(self, x, u):
        return float(np.linalg.norm(u))

    def proj(self, x, u):
        return u

    egrad2rgrad = proj

    def retr(self, x, u):
        return x + u

    exp = retr

    def transp(self, x1, x2, u):
        return u

    def rand(self):
        return np.random.randn(*self._shape)

    def zerovec(self, x):
        return np.zeros(self._shape)

    def dist(self, x, y):
        return float(np.linalg.norm(x - y))


class Sphere:
    """Unit sphere of R^n, points of shape (n,)."""

    def __init__(self, n):
        self._shape = (n,)
        self.dim = n - 1
        self.typicaldist = np.pi

    def inner(self, x, u, v):
        return float(np.dot(u.reshape(-1), v.reshape(-1)))

    def norm(self, x, u):
        return float(np.linalg.norm(u))

    def proj(self, x, u):
        return u - np.dot(x.reshape(-1), u.reshape(-1)) * x

    egrad2rgrad = proj

    def retr(self, x, u):
        y = x + u
        return y / np.linalg.norm(y)

    def transp(self, x1, x2, u):
        return self.proj(x2, u)

    def rand(self):
        y = np.random.randn(*self._shape)
        return y / np.linalg.norm(y)

    def zerovec(self, x):
        return np.zeros(self._shape)

    def dist(self, x, y):
        return float(np.arccos(np.clip(np.dot(x.reshape(-1), y.reshape(-1)), -1.0, 1.0)))


class Grassmann:
    """Subspaces of dimension p of R^n represented by orthonormal n x p matrices."""

    def __init__(self, n, p):
        self._n, self._p = n, p
        self._shape = (n, p)
        self.dim = n * p - p * p
        self.typicaldist = float(np.sqrt(p))

    def inner(self, x, u, v):
        return float(np.vdot(u, v))

    def norm(self, x, u):
        return float(np.linalg.norm(u))

    def proj(self, x, u):
        return u - x @ (x.T @ u)

    egrad2rgrad = proj

    def retr(self, x, u):
        # polar retraction: the orthonormal factor of x + u
        uu, _, vt = np.linalg.svd(x + u, full_matrices=False)
        return uu @ vt

    def transp(self, x1, x2, u):
        return self.proj(x2, u)

    def rand(self):
        q, _ = np.linalg.qr(np.random.randn(self._n, self._p))
        return q

    def zerovec(self, x):
        return np.zeros(self._shape)

    def dist(self, x, y):
        s = np.clip(np.linalg.svd(x.T @ y, compute_uv=False), -1.0, 1.0)      # cosines of the principal angles
        return float(np.linalg.norm(np.arccos(s)))


class PositiveDefinite:
    """S^n_++ with the affine-invariant metric, single matrices (hyper-parameter sized: the bottom block of the nested-SPD
    reconstruction) ([3P] pymanopt.manifolds.PositiveDefinite)."""

    def __init__(self, n):
        self._n = n
        self._shape = (n, n)
        self.dim = n * (n + 1) // 2
        self.typicaldist = float(np.sqrt(self.dim))
        self._inv = []                 # (bytes of x, x^-1) of the last base points: a solver asks for many inner products at one point

    @staticmethod
    def _sym(a):
        return 0.5 * (a + a.T)

    def _inverse(self, x):
        key = x.tobytes()
        for k, xi in self._inv:
            if k == key:
                return xi
        xi = np.linalg.inv(x)
        self._inv = [(key, xi)] + self._inv[:2]
        return xi

    def inner(self, x, u, v):
        xi = self._inverse(x)                                        # tr(X^-1 U X^-1 V)
        return float(np.vdot((xi @ u).T, xi @ v))

    def norm(self, x, u):
        return float(np.sqrt(max(self.inner(x, u, u), 0.0)))

    def proj(self, x, u):
        return self._sym(u)

    def egrad2rgrad(self, x, g):
        return x @ self._sym(g) @ x

    def retr(self, x, u):          # retr = exp = L expm(L^-1 U L^-T) L^T
        return self.retr_steps(x, u, (1.0,))[0]

    exp = retr

    def retr_steps(self, x, u, steps):
        """[retr(x, t u) for t in steps]: one Cholesky factor and one eigen-decomposition serve every step length (a line search asks for
        several along one direction): L expm(t S) L^T = (L Q) e^(t w) (L Q)^T with S = L^-1 U L^-T = Q diag(w) Q^T."""
        L = np.linalg.cholesky(x)
        Li = np.linalg.inv(L)
        w, q = np.linalg.eigh(self._sym(Li @ u @ Li.T))
        lq = L @ q
        return [self._sym((lq * np.exp(t * w)) @ lq.T) for t in steps]

    def transp(self, x1, x2, u):
        return u

    def rand(self):
        q, _ = np.linalg.qr(np.random.randn(self._n, self._n))
        return self._sym((q * (1.0 + np.random.rand(self._n))) @ q.T)

    def zerovec(self, x):
        return np.zeros(self._shape)

    def dist(self, x, y):
        L = np.linalg.cholesky(x)
        Li = np.linalg.inv(L)
        lam = np.linalg.eigvalsh(self._sym(Li @ y @ Li.T))
        return float(np.sqrt(np.sum(np.log(lam) ** 2)))


class PackedEuclideanSpheres:
    """A product of Euclidean and Sphere factors on ONE flat vector (the factors' entries one after the other): the same geometry as
    Product([...]) - inner products, projections, retractions and transports factor by factor - as a handful of whole-vector numpy
    operations (segment sums by np.add.reduceat) instead of a Python loop over the factors.  HD-GaBO on the sphere learns one axis per
    nested level (48 sphere factors at D = 51): with one launch per objective evaluation the loop over the factors was most of a fit."""

    def __init__(self, factors):
        if not factors or not all(type(m) in (Euclidean, Sphere) for m in factors):
            raise ValueError("PackedEuclideanSpheres takes Euclidean and Sphere factors only")
        self._factors = list(factors)
        sizes = [int(np.prod(m._shape)) for m in self._factors]
        self._bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self._starts = self._bounds[:-1]
        self._segment = np.repeat(np.arange(len(sizes)), sizes)
        self._sphere = np.array([type(m) is Sphere for m in self._factors])
        self.dim = int(sum(m.dim for m in self._factors))
        self.typicaldist = float(np.sqrt(sum(m.typicaldist ** 2 for m in self._factors)))

    def pack(self, parts):
        return np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in parts])

    def unpack(self, v):
        return [v[a:b].reshape(m._shape) for a, b, m in zip(self._bounds[:-1], self._bounds[1:], self._factors)]

    def _segment_sums(self, w):
        return np.add.reduceat(w, self._starts)

    def inner(self, x, u, v):
        return float(np.dot(u, v))

    def norm(self, x, u):
        return float(np.sqrt(max(np.dot(u, u), 0.0)))

    def proj(self, x, u):
        along = np.where(self._sphere, self._segment_sums(x * u), 0.0)
        return u - along[self._segment] * x

    egrad2rgrad = proj

    def retr(self, x, u):
        y = x + u
        scale = np.where(self._sphere, np.sqrt(self._segment_sums(y * y)), 1.0)
        return y / scale[self._segment]

    exp = retr

    def transp(self, x1, x2, u):
        return self.proj(x2, u)

    def rand(self):
        return self.pack([m.rand() for m in self._factors])

    def zerovec(self, x):
        return np.zeros(int(self._bounds[-1]))

    def dist(self, x, y):
        dots = self._segment_sums(x * y)
        diff = x - y
        sq = np.where(self._sphere, np.arccos(np.clip(dots, -1.0, 1.0)) ** 2, self._segment_sums(diff * diff))
        return float(np.sqrt(np.sum(sq)))


class Product:
    """Product manifold: points and tangent vectors are lists, one entry per factor."""

    def __init__(self, manifolds):
        self._manifolds = list(manifolds)
        self.dim = int(sum(m.dim for m in self._manifolds))
        self.typicaldist = float(np.sqrt(sum(m.typicaldist ** 2 for m in self._manifolds)))

    def _map(self, name, *args):
        return [getattr(m, name)(*[a[k] for a in args]) for k, m in enumerate(self._manifolds)]

    def inner(self, x, u, v):
        return float(sum(self._map("inner", x, u, v)))

    def norm(self, x, u):
        return float(np.sqrt(max(self.inner(x, u, u), 0.0)))

    def proj(self, x, u):
        return self._map("proj", x, u)

    def egrad2rgrad(self, x, u):
        return self._map("egrad2rgrad", x, u)

    def retr(self, x, u):
        return self._map("retr", x, u)

    def retr_steps(self, x, u, steps):
        """[retr(x, t u) for t in steps] (factors that can share work between the steps do: see PositiveDefinite.retr_steps)"""
        per = []
        for k, m in enumerate(self._manifolds):
            if hasattr(m, "retr_steps"):
                per.append(m.retr_steps(x[k], u[k], steps))
            else:
                per.append([m.retr(x[k], t * u[k]) for t in steps])
        return [[per[k][i] for k in range(len(self._manifolds))] for i in range(len(steps))]

    def transp(self, x1, x2, u):
        return self._map("transp", x1, x2, u)

    def rand(self):
        return [m.rand() for m in self._manifolds]

    def zerovec(self, x):
        return self._map("zerovec", x)

    def dist(self, x, y):
        return float(np.sqrt(sum(di * di for di in self._map("dist", x, y))))
