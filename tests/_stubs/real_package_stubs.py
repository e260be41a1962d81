"""Stand-ins with the SHAPE of the real gpytorch / botorch packages, injected into sys.modules BEFORE gabotorch_amd is imported, so that the
`try: import gpytorch` branches of gabotorch_amd._compat and gabotorch_amd.plugin_api run (that branch had never executed
anywhere - neither package exists in the image).  What is modelled is the base-class contract the reference's kernels rely on
(BoManifolds/kernel_utils/kernels_spd.py:33-70):

  gpytorch.Module        : torch.nn.Module with register_parameter / register_prior(name, prior, param_or_closure, setting_closure) /
                           register_constraint(param_name, constraint) -> attribute `<param_name>_constraint` / initialize(**kwargs)
  gpytorch.kernels.Kernel: `has_lengthscale` is a CLASS attribute (gpytorch >= 1.0; extra constructor keywords such as the reference's
                           has_lengthscale=False fall into **kwargs), raw_lengthscale + Positive constraint when set, batch_shape,
                           __call__(x1, x2=None, diag=False, **params) -> forward (dense here, a LazyTensor there)
  gpytorch.constraints   : GreaterThan(lower_bound) / Positive with transform / inverse_transform (softplus)
  botorch                : acquisition.ExpectedImprovement(model, best_f, maximize) as a torch Module calling model.posterior(X), models.SingleTaskGP,
                           fit_gpytorch_model
Only names and call signatures are imitated; nothing here is copied from either package."""
import sys
import types

import torch


def _inv_softplus(x):
    return x + torch.log(-torch.expm1(-x))


class Interval(torch.nn.Module):
    def __init__(self, lower_bound, upper_bound=float("inf"), transform=None, inv_transform=None, initial_value=None):
        super().__init__()
        self.register_buffer("lower_bound", torch.as_tensor(float(lower_bound)))
        self.initial_value = initial_value

    def transform(self, raw):
        return torch.nn.functional.softplus(raw) + self.lower_bound.to(raw)

    def inverse_transform(self, value):
        return _inv_softplus(value - self.lower_bound.to(value))


class GreaterThan(Interval):
    pass


class Positive(GreaterThan):
    def __init__(self, transform=None, inv_transform=None, initial_value=None):
        super().__init__(0.0, initial_value=initial_value)


class Module(torch.nn.Module):
    calls = []       # (class name, method, argument name): the record the test reads

    def __init__(self):
        super().__init__()
        self._priors = {}
        self._constraints = {}

    def register_parameter(self, name, parameter):
        Module.calls.append((type(self).__name__, "register_parameter", name))
        super().register_parameter(name, parameter)

    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        if not (isinstance(param_or_closure, str) or callable(param_or_closure)):
            raise TypeError("param_or_closure")
        Module.calls.append((type(self).__name__, "register_prior", name))
        self._priors[name] = (prior, param_or_closure, setting_closure)

    def register_constraint(self, param_name, constraint):
        if param_name not in self._parameters:
            raise RuntimeError("Attempting to register constraint for nonexistent parameter.")
        Module.calls.append((type(self).__name__, "register_constraint", param_name))
        self.add_module(param_name + "_constraint", constraint)
        self._constraints[param_name + "_constraint"] = constraint

    def initialize(self, **kwargs):
        for name, val in kwargs.items():
            param = getattr(self, name)
            with torch.no_grad():
                param.copy_(torch.as_tensor(val).to(param).expand_as(param))
        return self


class Kernel(Module):
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None, lengthscale_prior=None, lengthscale_constraint=None,
                 eps=1e-6, **kwargs):
        super().__init__()
        self._batch_shape = torch.Size(batch_shape)
        self.ard_num_dims = ard_num_dims
        self.active_dims = active_dims
        if self.has_lengthscale:
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*self._batch_shape, 1, 1 if ard_num_dims is None else ard_num_dims)))
            self.register_constraint("raw_lengthscale", lengthscale_constraint if lengthscale_constraint is not None else Positive())

    @property
    def batch_shape(self):
        return self._batch_shape

    @batch_shape.setter
    def batch_shape(self, value):
        self._batch_shape = torch.Size(value)

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_lengthscale)
        self.initialize(raw_lengthscale=self.raw_lengthscale_constraint.inverse_transform(value))

    def __call__(self, x1, x2=None, diag=False, last_dim_is_batch=False, **params):
        x1_ = x1.unsqueeze(-1) if x1.dim() == 1 else x1
        x2_ = x1_ if x2 is None else (x2.unsqueeze(-1) if x2.dim() == 1 else x2)
        return self.forward(x1_, x2_, diag=diag, **params)


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(tuple(self.batch_shape))))
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, lambda m=self: m.outputscale, lambda m, v: m._set_outputscale(v))
        self.register_constraint("raw_outputscale", outputscale_constraint if outputscale_constraint is not None else Positive())

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        self._set_outputscale(value)

    def _set_outputscale(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_outputscale)
        self.initialize(raw_outputscale=self.raw_outputscale_constraint.inverse_transform(value))

    def forward(self, x1, x2, diag=False, **params):
        k = self.base_kernel.forward(x1, x2, diag=diag, **params)
        return k * self.outputscale.to(k)


class GammaPrior:
    def __init__(self, concentration, rate):
        self.concentration, self.rate = float(concentration), float(rate)


class GaussianLikelihood:
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=None, **kwargs):
        self.noise_prior, self.noise_constraint = noise_prior, noise_constraint


class ExactMarginalLogLikelihood:
    def __init__(self, likelihood, model):
        self.likelihood, self.model = likelihood, model


def install():
    """puts `gpytorch` and `botorch` modules of this shape into sys.modules; returns them"""
    gp = types.ModuleType("gpytorch")
    gp.Module = Module
    gp.constraints = types.ModuleType("gpytorch.constraints")
    gp.constraints.GreaterThan, gp.constraints.Positive, gp.constraints.Interval = GreaterThan, Positive, Interval
    gp.kernels = types.ModuleType("gpytorch.kernels")
    gp.kernels.Kernel, gp.kernels.ScaleKernel = Kernel, ScaleKernel
    gp.likelihoods = types.ModuleType("gpytorch.likelihoods")
    gp.likelihoods.GaussianLikelihood = GaussianLikelihood
    gp.mlls = types.ModuleType("gpytorch.mlls")
    gp.mlls.ExactMarginalLogLikelihood = ExactMarginalLogLikelihood
    gp.priors = types.ModuleType("gpytorch.priors")
    gp.priors.GammaPrior = GammaPrior
    gp.priors.torch_priors = types.SimpleNamespace(GammaPrior=GammaPrior)
    for name in ("constraints", "kernels", "likelihoods", "mlls", "priors"):
        sys.modules["gpytorch." + name] = getattr(gp, name)
    sys.modules["gpytorch"] = gp

    bo = types.ModuleType("botorch")
    bo.acquisition = types.ModuleType("botorch.acquisition")
    bo.models = types.ModuleType("botorch.models")

    class AnalyticAcquisitionFunction(torch.nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

    class ExpectedImprovement(AnalyticAcquisitionFunction):
        """EI from model.posterior(X).mean / .variance  (botorch.acquisition.analytic [3P]): a FOREIGN acquisition object for the maximiser"""

        def __init__(self, model, best_f, maximize=True):
            super().__init__(model)
            self.best_f, self.maximize = float(best_f), maximize

        def forward(self, X):
            post = self.model.posterior(X)
            mean, var = (post.mean, post.variance) if hasattr(post, "mean") else post      # (this package's GP returns the pair)
            mean, sigma = mean.reshape(X.shape[:-2]), var.clamp_min(1e-18).sqrt().reshape(X.shape[:-2])
            u = (mean - self.best_f) / sigma
            u = u if self.maximize else -u
            normal = torch.distributions.Normal(torch.zeros_like(u), torch.ones_like(u))
            return sigma * (torch.exp(normal.log_prob(u)) + u * normal.cdf(u))

    bo.acquisition.ExpectedImprovement = ExpectedImprovement
    bo.acquisition.PosteriorMean = type("PosteriorMean", (AnalyticAcquisitionFunction,), {})
    bo.acquisition.analytic = types.SimpleNamespace(ExpectedImprovement=ExpectedImprovement, PosteriorMean=bo.acquisition.PosteriorMean)
    bo.models.SingleTaskGP = type("SingleTaskGP", (), {})
    bo.fit_gpytorch_model = lambda mll, **kw: mll
    sys.modules["botorch"], sys.modules["botorch.acquisition"], sys.modules["botorch.models"] = bo, bo.acquisition, bo.models
    return gp, bo
