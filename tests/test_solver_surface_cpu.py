"""The reference's solver surface (class names, constructor keywords, `solve(problem, x=ndarray, ...)`), on CPU:
TrustRegions / ConstrainedTrustRegions / StrictConstrainedTrustRegions drive pymanopt-style problems on single numpy points and
reproduce the optima of the reference's own solvers (tests/golden/trust_regions.npz); gen_candidates_manifold drives a foreign
pymanopt-style solver restart by restart."""
import numpy as np
import torch

from gabotorch_amd.manifold_optimization.approximate_hessian import get_hessianfd
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedProblem, BatchedTrustRegions
from gabotorch_amd.manifold_optimization.constrained_trust_regions import ConstrainedTrustRegions, StrictConstrainedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold
from gabotorch_amd.manifold_optimization.robust_trust_regions import TrustRegions
from gabotorch_amd.pymanopt_addons.problem import Problem
from tests._cpu_manifolds import CpuSphere, sphere_kernel_mean_cost

T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)   # noqa: E731


class NumpySphere:
    """pymanopt's duck type: single points / tangent vectors as numpy arrays"""

    def __init__(self, n):
        self._n, self._shape, self.dim, self.typicaldist = n, (n,), n - 1, np.pi
    inner = staticmethod(lambda x, u, v: float(np.dot(u, v)))
    norm = staticmethod(lambda x, u: float(np.linalg.norm(u)))
    proj = staticmethod(lambda x, h: h - np.dot(x, h) * x)
    egrad2rgrad = proj
    zerovec = staticmethod(np.zeros_like)

    def ehess2rhess(self, x, eg, eh, u):
        return self.proj(x, eh) - np.dot(x, eg) * u

    @staticmethod
    def retr(x, u):
        y = x + u
        return y / np.linalg.norm(y)

    def transp(self, x1, x2, d):
        return self.proj(x2, d)

    def rand(self):
        x = np.random.randn(self._n)
        return x / np.linalg.norm(x)


def test_reference_names_and_keywords():
    s = TrustRegions(mingradnorm=1e-5, maxiter=50, miniter=2, kappa=0.2, theta=0.9, rho_prime=0.05, rho_regularization=1e2)
    assert (s.mingradnorm, s.maxiter, s.miniter, s.kappa, s.theta, s.rho_prime, s.rho_regularization) == (1e-5, 50, 2, 0.2, 0.9, 0.05, 1e2)
    assert not s.strict_constraints and isinstance(s, BatchedTrustRegions)
    assert not ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100).strict_constraints            # examples/gabo_spd.py:183
    assert StrictConstrainedTrustRegions(mingradnorm=2e-4, maxiter=100, minstepsize=1e-4).strict_constraints   # hd_gabo_spd.py:194
    s._mingradnorm = 1e-3                                   # pymanopt-style outer solvers tighten the underscored attribute
    assert s.mingradnorm == 1e-3


def test_pymanopt_style_problem_matches_reference_trust_regions(golden):
    g = golden("trust_regions.npz")
    for n in (3, 5):
        batched_cost = sphere_kernel_mean_cost(T(g[f"sph{n}_Y"]), T(g[f"sph{n}_w"]), float(g[f"sph{n}_beta"]))
        cost = lambda x: batched_cost(x[None])[0]           # noqa: E731   one point -> 0-dim tensor
        for approx, key in ((False, "exact"), (True, "fd")):
            problem = Problem(manifold=NumpySphere(n), cost=cost, arg=torch.Tensor(), verbosity=0)
            if approx:
                import types
                problem._hess = types.MethodType(get_hessianfd, problem)            # manifold_optimize.py:202
            for i in range(3):
                x = TrustRegions().solve(problem, x=g[f"sph{n}_x0"][i])
                assert isinstance(x, np.ndarray) and x.shape == (n,)
                np.testing.assert_allclose(x, g[f"sph{n}_{key}_x"][i], rtol=0, atol=1e-6)
                np.testing.assert_allclose(problem.cost(x), g[f"sph{n}_{key}_f"][i], rtol=1e-9, atol=1e-12)
        # constrained, constraint as the reference passes it: a torch callable on one point returning a 0-dim tensor
        problem = Problem(manifold=NumpySphere(n), cost=cost, arg=torch.Tensor(), verbosity=0)
        con = lambda x: x[0] - 0.3                          # noqa: E731
        x = ConstrainedTrustRegions(mingradnorm=1e-6, maxiter=200).solve(problem, x=g[f"sph{n}_con_x0"][0], ineq_constraints=con)
        np.testing.assert_allclose(x, g[f"sph{n}_con_x"][0], rtol=0, atol=1e-6)
        x = StrictConstrainedTrustRegions(mingradnorm=1e-6, maxiter=200).solve(problem, x=g[f"sph{n}_con_x0"][0], ineq_constraints=[con])
        np.testing.assert_allclose(x, g[f"sph{n}_strict_x"][0], rtol=0, atol=1e-6)


def test_named_classes_equal_batched_solver_on_batched_problems(golden):
    g = golden("trust_regions.npz")
    cost = sphere_kernel_mean_cost(T(g["sph5_Y"]), T(g["sph5_w"]), float(g["sph5_beta"]))
    x0 = T(g["sph5_x0"])
    a = TrustRegions().solve(BatchedProblem(CpuSphere(5), cost), x0)
    b = BatchedTrustRegions().solve(BatchedProblem(CpuSphere(5), cost), x0)
    assert torch.equal(a, b)


def test_get_hessianfd_formula():
    rng = np.random.default_rng(0)
    n = 4
    Y = rng.standard_normal((6, n)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    batched_cost = sphere_kernel_mean_cost(T(Y), T(rng.uniform(0.2, 1, 6)), 1.1)
    problem = Problem(manifold=NumpySphere(n), cost=lambda x: batched_cost(x[None])[0], arg=torch.Tensor(), verbosity=0)
    x = rng.standard_normal(n); x /= np.linalg.norm(x)
    a = NumpySphere.proj(x, rng.standard_normal(n))
    exact = problem.hess(x, a)
    fd = get_hessianfd(problem, x, a)
    np.testing.assert_allclose(fd, exact, rtol=0, atol=5e-4 * np.linalg.norm(exact))      # O(eps |a|) truncation, eps = 2^-14
    np.testing.assert_array_equal(get_hessianfd(problem, x, np.zeros(n)), np.zeros(n))       # |a| < 1e-15 (:36-37)


class _GradientDescent:
    """a foreign solver with pymanopt's interface: solve(problem, x=ndarray)"""

    def __init__(self):
        self.calls = 0

    def solve(self, problem, x=None, eq_constraints=None, ineq_constraints=None):
        self.calls += 1
        man = problem.manifold
        for _ in range(300):
            g = problem.grad(x)
            if man.norm(x, g) < 1e-9:
                break
            x = man.retr(x, -0.5 * g)
        return x


def test_gen_candidates_drives_a_foreign_solver_restart_by_restart():
    rng = np.random.default_rng(1)
    n = 3
    y = rng.standard_normal(n); y /= np.linalg.norm(y)

    def acq(X):                                  # b x 1 x n -> b (or q=1 x n -> 1, botorch's t-batch convention): maximal at y
        X = X if X.dim() == 3 else X[None]
        return (X[:, 0] * T(y)).sum(-1)

    x0 = rng.standard_normal((4, 1, n)); x0 /= np.linalg.norm(x0, axis=-1, keepdims=True)
    solver = _GradientDescent()
    cand, val = gen_candidates_manifold(T(x0), acq, NumpySphere(n), solver)
    assert solver.calls == 4 and cand.shape == (4, 1, n) and val.shape == (4,)
    np.testing.assert_allclose(cand[:, 0].numpy(), np.tile(y, (4, 1)), atol=1e-7)
    np.testing.assert_allclose(val.numpy(), 1.0, atol=1e-12)


def test_use_rand_start_reaches_the_same_optima(golden):
    """use_rand=True (robust_trust_regions.py:173-219, 407-452): random tCG start without preconditioner + Cauchy-point safeguard.  The
    random start cannot be pinned against the reference (pymanopt's randvec and numpy's global stream), so the property is checked:
    every restart ends on a stationary point at least as good as its start, and from starts near an optimum it ends on that optimum."""
    g = golden("trust_regions.npz")
    cost = sphere_kernel_mean_cost(T(g["sph5_Y"]), T(g["sph5_w"]), float(g["sph5_beta"]))
    torch.manual_seed(0)
    x0 = T(g["sph5_x0"])
    prob = BatchedProblem(CpuSphere(5), cost)
    x = TrustRegions(use_rand=True).solve(prob, x0)
    assert (cost(x) <= cost(x0) + 1e-12).all()
    _, rg = prob.cost_grad(x)
    assert float(rg.norm(dim=-1).max()) < 1e-5
    near = T(g["sph5_exact_x"]) + 1e-3 * torch.randn(g["sph5_exact_x"].shape, dtype=torch.float64)
    near = near / near.norm(dim=-1, keepdim=True)
    y = TrustRegions(use_rand=True).solve(BatchedProblem(CpuSphere(5), cost), near)
    np.testing.assert_allclose(y.numpy(), g["sph5_exact_x"], atol=1e-6)


def test_plugin_api_namespaces_expose_the_names_the_examples_import():
    from gabotorch_amd.plugin_api import botorch, gpytorch
    import gabotorch_amd.plugin_api.pymanopt.manifolds as pyman_man
    import gabotorch_amd.plugin_api.pymanopt.solvers as pyman_solvers
    for obj in (gpytorch.kernels.ScaleKernel, gpytorch.kernels.Kernel, gpytorch.priors.torch_priors.GammaPrior,
                gpytorch.likelihoods.gaussian_likelihood.GaussianLikelihood, gpytorch.constraints.GreaterThan,
                gpytorch.mlls.ExactMarginalLogLikelihood, botorch.models.SingleTaskGP, botorch.fit_gpytorch_model,
                botorch.acquisition.ExpectedImprovement, pyman_man.PositiveDefinite, pyman_man.Sphere, pyman_man.Grassmann,
                pyman_man.Product, pyman_man.Euclidean, pyman_solvers.TrustRegions, pyman_solvers.ConjugateGradient):
        assert callable(obj)
    prior = gpytorch.priors.torch_priors.GammaPrior(1.1, 0.05)
    assert abs((prior.concentration - 1) / prior.rate - 2.0) < 1e-12                     # examples/gabo_spd.py:168
    lik = gpytorch.likelihoods.gaussian_likelihood.GaussianLikelihood(noise_prior=prior, noise_constraint=gpytorch.constraints.GreaterThan(1e-8),
                                                                     initial_value=2.0)
    assert lik.initial_value == 2.0


def test_one_point_constraint_callables_when_restarts_equal_dimension(golden):
    """The reference's constraint callables take ONE point (`x[1] - yc`, gabo_sphere_equality_constraints.py:106-107).  Applied to the
    R x dim batch of the lock-step solver such a callable returns row 1 - a vector of length dim, which has the batch's shape exactly when
    the number of restarts equals the dimension (5 restarts on S^4: the examples' num_restarts with dim = 5 of their beta_min ladder).  A shape
    test alone accepted that until round 5; the batched call is now also compared with the single-point call at both ends of the batch."""
    g = golden("tr_traces.npz")
    n = 5
    Y, w, beta = T_(g["sph5_Y"]), T_(g["sph5_w"]), float(g["sph5_beta"])
    rng = np.random.default_rng(8)
    x0 = rng.standard_normal((n, n))                 # R = dim = 5
    x0[:, 1] = 0.0
    x0 /= np.linalg.norm(x0, axis=1, keepdims=True)
    out = {}
    for name, con in (("one_point", lambda x: x[1] - 0.0), ("batch_safe", lambda x: x[..., 1] - 0.0)):
        prob = BatchedProblem(CpuSphere(n), sphere_kernel_mean_cost(Y, w, beta), approx_hessian=False)
        solver = ConstrainedTrustRegions(mingradnorm=1e-6, maxiter=25)
        x = solver.solve(prob, T_(x0), eq_constraints=[con])
        out[name] = (x.numpy(), solver.log["per_restart_iterations"].numpy())
        vals = BatchedTrustRegions._constraint_values(T_(x0 + 0.01), [con]).numpy()[:, 0]
        np.testing.assert_allclose(vals, (x0 + 0.01)[:, 1], rtol=0, atol=1e-15)          # each restart's OWN coordinate 1
    np.testing.assert_array_equal(out["one_point"][1], out["batch_safe"][1])
    np.testing.assert_allclose(out["one_point"][0], out["batch_safe"][0], rtol=0, atol=1e-12)
    assert np.abs(out["one_point"][0][:, 1]).max() < 1e-5                                 # the equality holds at the end (to Delta_cons)


def T_(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)
