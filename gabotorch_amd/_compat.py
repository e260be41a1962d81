"""gpytorch plugin surface.  When gpytorch is importable the kernels subclass the real `gpytorch.kernels.Kernel`; when it
is not (this image has no gpytorch and no network) a minimal stand-in with the same registration API is used so the
kernel classes keep their reference signatures.  The stand-in only covers what the reference kernels call
(kernels_spd.py:33-70): has_lengthscale / batch_shape kwargs, register_parameter, register_prior, register_constraint,
`raw_<name>_constraint`, initialize(), and a dense __call__.
"""

import torch

try:  # (exercised by tests/test_real_package_branch_cpu.py with modules of the real package's shape)
    import gpytorch
    from gpytorch.constraints import GreaterThan, Positive
    from gpytorch.kernels import Kernel, ScaleKernel
    HAVE_GPYTORCH = True
except Exception:  # noqa: BLE001
    gpytorch = None
    HAVE_GPYTORCH = False

    def _inv_softplus(x):
        return x + torch.log(-torch.expm1(-x))

    class GreaterThan(torch.nn.Module):
        """softplus(raw) + lower_bound   (gpytorch.constraints.GreaterThan semantics)."""

        def __init__(self, lower_bound, transform=None, inv_transform=None, initial_value=None):
            super().__init__()
            self.lower_bound = torch.as_tensor(float(lower_bound))
            self.initial_value = initial_value

        def transform(self, raw):
            return torch.nn.functional.softplus(raw) + self.lower_bound.to(raw)

        def inverse_transform(self, value):
            return _inv_softplus(value - self.lower_bound.to(value))

    class Positive(GreaterThan):
        def __init__(self, **kw):
            super().__init__(0.0, **kw)

    class Kernel(torch.nn.Module):
        def __init__(self, has_lengthscale=False, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None, **kwargs):
            super().__init__()
            self.has_lengthscale = has_lengthscale
            self.batch_shape = torch.Size(batch_shape)
            self.active_dims = active_dims
            self._priors = {}
            if has_lengthscale:     # gpytorch.kernels.Kernel: raw_lengthscale (*batch, 1, 1) under a Positive constraint
                self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, 1)))
                self.register_constraint("raw_lengthscale", kwargs.get("lengthscale_constraint") or Positive())

        @property
        def lengthscale(self):
            return self.raw_lengthscale_constraint.transform(self.raw_lengthscale) if self.has_lengthscale else None

        @lengthscale.setter
        def lengthscale(self, value):
            if not torch.is_tensor(value):
                value = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype, device=self.raw_lengthscale.device)
            self.initialize(raw_lengthscale=self.raw_lengthscale_constraint.inverse_transform(value))

        def register_parameter(self, name, parameter):
            super().register_parameter(name, parameter)

        def register_constraint(self, param_name, constraint):
            self.add_module(param_name + "_constraint", constraint)

        def register_prior(self, name, prior, param_or_closure, setting_closure=None):
            self._priors[name] = (prior, param_or_closure, setting_closure)

        def initialize(self, **kwargs):
            for name, val in kwargs.items():
                param = getattr(self, name)
                with torch.no_grad():
                    param.copy_(torch.as_tensor(val).to(param).expand_as(param))
            return self

        def __call__(self, x1, x2=None, diag=False, **params):
            x2 = x1 if x2 is None else x2
            if x1.dim() == 1:
                x1 = x1.unsqueeze(-1)
            if x2.dim() == 1:
                x2 = x2.unsqueeze(-1)
            res = self.forward(x1, x2, diag=diag, **params)
            if diag and res.shape[-2:] == (x1.shape[-2], x2.shape[-2]) and x1.shape[-2] == x2.shape[-2] and x1.shape[-2] != 1:
                res = res.diagonal(dim1=-2, dim2=-1)      # a kernel that ignored `diag` returns the full matrix
            return res

    class ScaleKernel(Kernel):
        """outputscale * base_kernel   (gpytorch.kernels.ScaleKernel semantics, dense)."""

        def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
            super().__init__(**kwargs)
            self.base_kernel = base_kernel
            self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(tuple(self.batch_shape))))
            self.register_constraint("raw_outputscale", outputscale_constraint or Positive())
            if outputscale_prior is not None:
                self.register_prior("outputscale_prior", outputscale_prior, lambda: self.outputscale,
                                    lambda v: self._set_outputscale(v))

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
