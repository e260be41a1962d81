"""The plugin surface seen from a reference example: the statements of examples/bo_spd/benchmark_examples/gabo_spd.py:79-220 (data,
kernel, likelihood, SingleTaskGP, mll, ConstrainedTrustRegions, fit_gpytorch_model, ExpectedImprovement, joint_optimize_manifold,
set_train_data) executed with nothing but the import prefixes changed - `BoManifolds.` -> `gabotorch_amd.`, and gpytorch / botorch /
pymanopt taken from gabotorch_amd.plugin_api (those packages are not installed here).  Data lives on the CPU as in the example; the
kernels, their gradients, the surrogate fit and the acquisition sweep run on the GPU."""
import functools
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gabo_spd_example_call_sequence():
    from gabotorch_amd.plugin_api import botorch, gpytorch
    import gabotorch_amd.plugin_api.pymanopt.manifolds as pyman_man
    from gabotorch_amd.Riemannian_utils.spd_utils import symmetric_matrix_to_vector_mandel, spd_sample
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,
                                                                vector_to_symmetric_matrix_mandel_torch)
    from gabotorch_amd.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch
    from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
    from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
    from gabotorch_amd.manifold_optimization.constrained_trust_regions import ConstrainedTrustRegions
    from gabotorch_amd.BO_test_functions.test_functions_spd import ackley_function_spd, optimum_ackley_spd

    np.random.seed(1234)
    torch.manual_seed(1234)
    dim = 3
    dim_vec = int(dim + dim * (dim - 1) / 2)
    spd_manifold = pyman_man.PositiveDefinite(dim)
    spd_manifold.rand = types.MethodType(spd_sample, spd_manifold)
    test_function = functools.partial(ackley_function_spd, spd_manifold=spd_manifold)
    true_min, true_opt_val = optimum_ackley_spd(spd_manifold)
    assert abs(float(true_opt_val)) < 1e-12
    min_eigenvalue, max_eigenvalue = 0.001, 5.0
    spd_manifold.min_eig, spd_manifold.max_eig = min_eigenvalue, max_eigenvalue
    lower_bound = torch.cat((min_eigenvalue * torch.ones(dim, dtype=torch.float64),
                             -max_eigenvalue / np.sqrt(2) * torch.ones(dim_vec - dim, dtype=torch.float64)))
    upper_bound = torch.cat((max_eigenvalue * torch.ones(dim, dtype=torch.float64),
                             max_eigenvalue / np.sqrt(2) * torch.ones(dim_vec - dim, dtype=torch.float64)))
    bounds = torch.stack([lower_bound, upper_bound])
    inequality_constraints = [functools.partial(max_eigenvalue_constraint_torch, maximum_eigenvalue=max_eigenvalue)]
    nb_data_init = 5
    x_data = torch.tensor(np.array([symmetric_matrix_to_vector_mandel(spd_manifold.rand()) for _ in range(nb_data_init)]))
    y_data = torch.zeros(nb_data_init, dtype=torch.float64)
    for n in range(nb_data_init):
        y_data[n] = test_function(x_data[n])
    k_fct = gpytorch.kernels.ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=0.5),
                                         outputscale_prior=gpytorch.priors.torch_priors.GammaPrior(2.0, 0.15))
    noise_prior = gpytorch.priors.torch_priors.GammaPrior(1.1, 0.05)
    noise_prior_mode = (noise_prior.concentration - 1) / noise_prior.rate
    lik_fct = gpytorch.likelihoods.gaussian_likelihood.GaussianLikelihood(
        noise_prior=noise_prior, noise_constraint=gpytorch.constraints.GreaterThan(1e-8), initial_value=noise_prior_mode)
    model = botorch.models.SingleTaskGP(x_data, y_data[:, None], covar_module=k_fct, likelihood=lik_fct)
    mll_fct = gpytorch.mlls.ExactMarginalLogLikelihood(model.likelihood, model)
    solver = ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100)
    new_best_f, index = y_data.min(0)
    best_x, best_f = [x_data[index]], [new_best_f]
    for iteration in range(4):
        botorch.fit_gpytorch_model(mll=mll_fct)
        acq_fct = botorch.acquisition.ExpectedImprovement(model=model, best_f=best_f[-1], maximize=False)
        new_x = joint_optimize_manifold(acq_fct, spd_manifold, solver, q=1, num_restarts=5, raw_samples=100, bounds=bounds,
                                        pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                        post_processing_manifold=symmetric_matrix_to_vector_mandel_torch,
                                        approx_hessian=True, inequality_constraints=inequality_constraints)
        assert new_x.shape == (1, dim_vec)
        new_y = test_function(new_x)[0]
        x_data = torch.cat((x_data, new_x.to(x_data)))
        y_data = torch.cat((y_data, new_y.to(y_data)))
        new_best_f, index = y_data.min(0)
        best_x.append(x_data[index])
        best_f.append(new_best_f)
        model.set_train_data(x_data, y_data, strict=False)
    assert x_data.shape == (nb_data_init + 4, dim_vec) and torch.isfinite(y_data).all()
    mats = vector_to_symmetric_matrix_mandel_torch(x_data).cpu().numpy()
    lam = np.linalg.eigvalsh(mats)
    assert lam.min() > 0 and lam.max() < max_eigenvalue + 0.5                    # the reference's constraint handling is soft
    assert all(float(b) <= float(a) + 1e-12 for a, b in zip(best_f, best_f[1:]))


def test_foreign_pymanopt_style_solver_runs_the_sphere_sweep():
    """joint_optimize_manifold with a solver that is NOT one of this package's classes (duck type `solve(problem, x=ndarray)`,
    manifold_optimize.py:211-220): restarts are driven one by one, the kernel evaluations inside the acquisition are still the HIP path."""
    from gabotorch_amd import manifolds, models
    from gabotorch_amd._compat import ScaleKernel
    from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel
    from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
    from gabotorch_amd.manifold_optimization.robust_trust_regions import TrustRegions

    class Foreign:
        def __init__(self):
            self.inner, self.calls = TrustRegions(mingradnorm=1e-6, maxiter=50), 0

        def solve(self, problem, x=None):
            self.calls += 1
            assert isinstance(x, np.ndarray) and x.shape == (3,)
            assert isinstance(problem.cost(x), float) and problem.grad(x).shape == (3,)
            return self.inner.solve(problem, x=x)

    rng = np.random.default_rng(3)
    np.random.seed(3)
    torch.manual_seed(3)
    xs = rng.standard_normal((12, 3)); xs /= np.linalg.norm(xs, axis=1, keepdims=True)
    X = torch.tensor(xs, device="cuda:0")
    y = torch.tensor(np.sin(3 * xs[:, 0]) + xs[:, 1] ** 2, device="cuda:0")
    gp = models.SingleTaskGP(X, y, ScaleKernel(SphereGaussianKernel(beta_min=6.5)), noise_prior=models.GammaPrior(1.1, 0.05))
    models.fit_gpytorch_model(gp)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    man = manifolds.Sphere(3)
    foreign = Foreign()
    np.random.seed(5); torch.manual_seed(5)
    a = joint_optimize_manifold(acq, man, foreign, q=1, num_restarts=3, raw_samples=40, bounds=None, options={"device": "cuda:0"})
    np.random.seed(5); torch.manual_seed(5)
    b = joint_optimize_manifold(acq, man, TrustRegions(mingradnorm=1e-6, maxiter=50), q=1, num_restarts=3, raw_samples=40, bounds=None,
                                options={"device": "cuda:0"})
    assert foreign.calls == 3 and a.shape == (1, 3)
    np.testing.assert_allclose(float(a.norm()), 1.0, atol=1e-12)
    # the lock-step device path and the one-by-one host-driven path reach the same maximiser from the same initial conditions
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5)
