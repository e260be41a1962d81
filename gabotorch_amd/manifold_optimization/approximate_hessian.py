"""`get_hessianfd`: finite-difference Riemannian Hessian-vector product, bound onto a problem as its `_hess`
(BoManifolds/manifold_optimization/approximate_hessian.py:11-62; manifold_optimize.py:198-202 does
`problem._hess = types.MethodType(get_hessianfd, problem)`).

For the library's batched problems the same formula runs inside `BatchedProblem.hess` / the HIP trust-region kernels; this function is
the single-point form for pymanopt-style problems."""
import numpy as np


def get_hessianfd(self, x, a):
    """(transp_{x1 -> x} grad f(x1) - grad f(x)) / c   with   x1 = retr_x(c a),  c = 2^-14 / |a|_x ;  zero for |a|_x < 1e-15."""
    man = self.manifold
    norm_a = man.norm(x, a)
    g0 = self.grad(x)
    sequence = isinstance(x, (list, tuple))
    if norm_a < 1e-15:
        return [np.zeros(np.shape(g)) for g in g0] if sequence else np.zeros(np.shape(g0))
    c = 2.0 ** -14 / norm_a
    x1 = man.retr(x, [c * ai for ai in a] if sequence else c * a)
    g1 = man.transp(x1, x, self.grad(x1))
    if sequence:                                                        # product manifolds hand lists around (:53-58)
        return [g1k / c - g0k / c for g1k, g0k in zip(g1, g0)]
    return g1 / c - g0 / c
