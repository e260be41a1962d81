"""Host-side pieces of the surrogate fit (manifold_gp_fit.py in the reference): numpy manifolds and Riemannian conjugate gradients.
No GPU needed: they only ever move hyper-parameters."""
import numpy as np

from gabotorch_amd.manifold_optimization.conjugate_gradient import ConjugateGradient
from gabotorch_amd.manifold_optimization.host_manifolds import Euclidean, Grassmann, Product, Sphere


class _Problem:
    def __init__(self, manifold, cost, egrad):
        self.manifold, self.cost, self._egrad = manifold, cost, egrad

    def grad(self, x):
        return self.manifold.egrad2rgrad(x, self._egrad(x))


def test_manifold_axioms():
    np.random.seed(0)
    for man in (Sphere(6), Grassmann(7, 3), Euclidean(4)):
        x = man.rand()
        u = man.proj(x, np.random.randn(*man._shape))
        v = man.proj(x, np.random.randn(*man._shape))
        np.testing.assert_allclose(man.proj(x, u), u, atol=1e-14)                       # projection is idempotent
        y = man.retr(x, 0.3 * u)
        if isinstance(man, Sphere):
            assert abs(np.linalg.norm(y) - 1) < 1e-14 and abs(np.dot(x, u)) < 1e-14
        if isinstance(man, Grassmann):
            np.testing.assert_allclose(y.T @ y, np.eye(3), atol=1e-13)
            np.testing.assert_allclose(x.T @ u, 0, atol=1e-13)
        w = man.transp(x, y, v)
        np.testing.assert_allclose(man.proj(y, w), w, atol=1e-13)                       # transported vector is tangent at y
        assert abs(man.inner(x, u, v) - man.inner(x, v, u)) < 1e-14
        # first-order retraction: retr(x, t u) = x + t u + O(t^2)
        t = 1e-6
        np.testing.assert_allclose((man.retr(x, t * u) - x) / t, u, atol=1e-5)


def test_conjugate_gradient_rayleigh_quotient_on_sphere():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((8, 8))
    a = a + a.T
    man = Sphere(8)
    np.random.seed(1)
    x, log = ConjugateGradient(maxiter=500, mingradnorm=1e-9).solve(_Problem(man, lambda x: float(x @ a @ x), lambda x: 2 * a @ x))
    lam = np.linalg.eigvalsh(a)
    assert abs(log["final_cost"] - lam[0]) < 1e-9 and abs(np.linalg.norm(x) - 1) < 1e-12
    assert all(b <= a_ + 1e-12 for a_, b in zip(log["cost_history"], log["cost_history"][1:]))      # monotone decrease


def test_conjugate_gradient_dominant_subspace_on_product():
    rng = np.random.default_rng(2)
    a = rng.standard_normal((9, 9))
    a = a @ a.T
    target = rng.standard_normal(3)
    man = Product([Grassmann(9, 2), Euclidean(3)])
    cost = lambda x: float(-np.trace(x[0].T @ a @ x[0]) + np.sum((x[1] - target) ** 2))            # noqa: E731
    egrad = lambda x: [-2 * a @ x[0], 2 * (x[1] - target)]                                           # noqa: E731
    np.random.seed(2)
    x, log = ConjugateGradient(maxiter=2000, mingradnorm=1e-8).solve(_Problem(man, cost, egrad))
    lam = np.linalg.eigvalsh(a)
    assert abs(-log["final_cost"] - (lam[-1] + lam[-2])) < 1e-6
    np.testing.assert_allclose(x[1], target, atol=1e-6)
    np.testing.assert_allclose(x[0].T @ x[0], np.eye(2), atol=1e-12)


def test_host_positive_definite_manifold():
    from gabotorch_amd.manifold_optimization.host_manifolds import PositiveDefinite
    np.random.seed(3)
    man = PositiveDefinite(4)
    x, y = man.rand(), man.rand()
    assert np.linalg.eigvalsh(x).min() > 0
    u = man.proj(x, np.random.randn(4, 4))
    np.testing.assert_allclose(u, u.T)
    z = man.retr(x, 0.3 * u)
    assert np.linalg.eigvalsh(z).min() > 0
    np.testing.assert_allclose(man.dist(x, z), 0.3 * man.norm(x, u), rtol=1e-10)      # exp is a geodesic: d(x, exp_x(t u)) = t |u|_x
    np.testing.assert_allclose(man.dist(x, y), man.dist(y, x), rtol=1e-12)
    a = np.random.randn(4, 4)
    np.testing.assert_allclose(man.dist(a @ x @ a.T, a @ y @ a.T), man.dist(x, y), rtol=1e-9)      # affine invariance
    g = np.random.randn(4, 4)
    rg = man.egrad2rgrad(x, g)
    np.testing.assert_allclose(man.inner(x, rg, u), np.tensordot(0.5 * (g + g.T), u, axes=2), rtol=1e-10)   # <rgrad, u>_x = <egrad, u>


def test_augmented_lagrange_method_on_a_constrained_toy_problem():
    """min x^T A x over R^5 subject to |x|^2 = 1 (equality) and x_0 >= 0.1 (inequality, inactive at the solution): lambda_min(A)."""
    from gabotorch_amd.manifold_optimization.augmented_lagrange_method import AugmentedLagrangeMethod, _Constraint
    rng = np.random.default_rng(4)
    a = rng.standard_normal((5, 5))
    a = a @ a.T + np.eye(5)
    man = Product([Euclidean(5)])

    class P:
        manifold = man
        cost = staticmethod(lambda x: float(x[0] @ a @ x[0]))
        grad = staticmethod(lambda x: [2 * a @ x[0]])
    eq = _Constraint(man, lambda x: (float(x[0] @ x[0] - 1.0), [2 * x[0]]))
    ineq = _Constraint(man, lambda x: (float(x[0][0] - 0.1), [np.eye(5)[0]]))
    np.random.seed(4)
    solver = AugmentedLagrangeMethod(ConjugateGradient(maxiter=300), maxiter=60, ending_tolgradnorm=1e-8)
    x = solver.solve(P, x=[np.ones(5) / np.sqrt(5)], eq_constraints=[eq], ineq_constraints=[ineq])[0]
    lam, vec = np.linalg.eigh(a)
    assert abs(vec[0, 0]) > 0.15            # the inequality is inactive at the global solution: the optimum is the smallest eigenvalue
    # the method stops when the iterate moves less than 1e-10 (pymanopt's default minstepsize), with the violation at ~1e-4
    assert abs(x @ x - 1.0) < 1e-3 and x[0] >= 0.1 - 1e-3
    np.testing.assert_allclose(x @ a @ x, lam[0], rtol=2e-3)


def test_library_constraints_are_recognised_for_graph_capture():
    """functools.partial over the library's eigenvalue constraints (plain or nested) is what the maximiser may capture with the
    trust-region graphs; anything else (lambdas, partials over user functions) stays eager."""
    import functools
    from gabotorch_amd.manifold_optimization.manifold_optimize import _library_constraint
    from gabotorch_amd.nested_mappings.nested_spd_constraints_utils import max_eigenvalue_nested_spd_constraint
    from gabotorch_amd.Riemannian_utils.spd_constraints_utils_torch import builtin_constraint, min_eigenvalue_constraint_torch
    plain = functools.partial(min_eigenvalue_constraint_torch, minimum_eigenvalue=0.1)
    nested = functools.partial(max_eigenvalue_nested_spd_constraint, maximum_eigenvalue=5.0, projection_matrix=None,
                               projection_complement_matrix=None, bottom_spd_matrix=None, contraction_matrix=None)
    assert _library_constraint(plain) and _library_constraint(nested)
    assert not _library_constraint(lambda x: min_eigenvalue_constraint_torch(x, 0.1))
    assert not _library_constraint(functools.partial(lambda x, b: x.sum() - b, b=1.0))
    assert builtin_constraint(plain) is not None and builtin_constraint(nested) is None      # only the plain ones run inside the kernel
