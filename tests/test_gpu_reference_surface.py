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


def test_hd_gabo_objectives_compose_projection_and_test_function():
    """examples/hd_bo_spd/benchmark_examples/hd_gabo_spd.py builds its objective as functools.partial(projected_function_spd, ...) and
    hd_gabo_sphere.py as functools.partial(nested_function_sphere, ...): the value is the latent test function at the projected point."""
    import gabotorch_amd.plugin_api.pymanopt.manifolds as pyman_man
    from gabotorch_amd.BO_test_functions.nested_test_functions_spd import projected_function_spd, optimum_projected_function_spd
    from gabotorch_amd.BO_test_functions.nested_test_functions_sphere import nested_function_sphere, optimum_nested_function_sphere
    from gabotorch_amd.BO_test_functions.test_functions_spd import rosenbrock_function_spd, optimum_rosenbrock_spd
    from gabotorch_amd.BO_test_functions.test_functions_sphere import ackley_function_sphere, optimum_ackley_sphere
    from gabotorch_amd.nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere
    from oracle import spd as ospd

    rng = np.random.default_rng(5)
    D, d = 6, 2
    q = np.linalg.qr(rng.standard_normal((D, D)))[0]
    x = q @ np.diag(rng.uniform(0.3, 3.0, D)) @ q.T
    x = 0.5 * (x + x.T)
    w = np.linalg.qr(rng.standard_normal((D, d)))[0]
    latent_manifold = pyman_man.PositiveDefinite(d)
    objective = functools.partial(projected_function_spd, low_dimensional_spd_manifold=latent_manifold, test_function=rosenbrock_function_spd,
                                  projection_matrix=torch.tensor(w, device="cuda"))
    xv = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(x[None]), device="cuda")
    got = objective(xv)
    assert tuple(got.shape) == (1, 1)
    want = rosenbrock_function_spd(torch.tensor(ospd.symmetric_matrix_to_vector_mandel((w.T @ x @ w)[None])), latent_manifold)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-11)
    opt_x, opt_y = optimum_projected_function_spd(optimum_rosenbrock_spd, latent_manifold, w)
    ref_x, ref_y = optimum_rosenbrock_spd(latent_manifold)
    np.testing.assert_array_equal(np.asarray(opt_x), np.asarray(ref_x))
    assert abs(float(np.asarray(opt_y).reshape(-1)[0])) < 1e-12                 # Rosenbrock's minimum value

    dim, sub = 5, 3                                                             # S^4 -> S^2, two nested axes
    axes, dists = [], []
    for k in range(dim - sub):
        a = rng.standard_normal(dim - k)
        axes.append(torch.tensor(a / np.linalg.norm(a), device="cuda"))
        dists.append(torch.tensor(float(rng.uniform(1.0, 1.4)), dtype=torch.float64, device="cuda"))
    sub_manifold = pyman_man.Sphere(sub)
    p = rng.standard_normal((1, dim))
    p = torch.tensor(p / np.linalg.norm(p), device="cuda")
    got = nested_function_sphere(p[0], sub_manifold, ackley_function_sphere, axes, dists)
    want = ackley_function_sphere(projection_from_sphere_to_subsphere(p, axes, dists)[-1], sub_manifold)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-12)
    opt_x, opt_y = optimum_nested_function_sphere(optimum_ackley_sphere, sub_manifold, axes, dists)
    assert opt_x.shape == (1, dim) and abs(np.linalg.norm(opt_x) - 1.0) < 1e-12
    back = nested_function_sphere(torch.tensor(opt_x, device="cuda"), sub_manifold, ackley_function_sphere, axes, dists)
    np.testing.assert_allclose(back.cpu().numpy(), np.asarray(opt_y), atol=1e-7)   # the lifted optimum projects back onto the subsphere's optimum
