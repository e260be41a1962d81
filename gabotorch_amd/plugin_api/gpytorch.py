"""`gpytorch` as the reference examples use it."""
import types as _types

try:                                     # (tests/test_real_package_branch_cpu.py)
    from gpytorch import constraints, kernels, likelihoods, mlls, priors  # noqa: F401
except Exception:                        # noqa: BLE001
    from .. import _compat, models

    class GaussianLikelihood:
        """gpytorch.likelihoods.GaussianLikelihood(noise_prior=, noise_constraint=, initial_value=) [3P]: a homoskedastic noise
        description consumed by SingleTaskGP (examples/gabo_spd.py:169-173)."""

        def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=None, initial_value=None, **kwargs):
            self.noise_prior = noise_prior
            self.noise_constraint = noise_constraint
            self.initial_value = initial_value

    class ExactMarginalLogLikelihood:
        """gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model) [3P]: the handle `fit_gpytorch_model(mll=...)` takes"""

        def __init__(self, likelihood, model):
            self.likelihood, self.model = likelihood, model

        def __call__(self, *args, **kwargs):
            return self.model.marginal_log_likelihood()

    _torch_priors = _types.SimpleNamespace(GammaPrior=models.GammaPrior)
    kernels = _types.SimpleNamespace(Kernel=_compat.Kernel, ScaleKernel=_compat.ScaleKernel)
    priors = _types.SimpleNamespace(GammaPrior=models.GammaPrior, torch_priors=_torch_priors)
    constraints = _types.SimpleNamespace(GreaterThan=_compat.GreaterThan, Positive=_compat.Positive)
    likelihoods = _types.SimpleNamespace(GaussianLikelihood=GaussianLikelihood,
                                         gaussian_likelihood=_types.SimpleNamespace(GaussianLikelihood=GaussianLikelihood))
    mlls = _types.SimpleNamespace(ExactMarginalLogLikelihood=ExactMarginalLogLikelihood)
