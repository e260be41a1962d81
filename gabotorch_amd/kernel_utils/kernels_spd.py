"""GPyTorch kernels on the SPD manifold, computed by libgabo_hip.so on the MI355X.

Same class names, constructor arguments, `beta` parameterisation and `forward` signatures as the reference
(BoManifolds/kernel_utils/kernels_spd.py:17-187).  Inputs are Mandel vectors (..., N, d(d+1)/2); the Mandel->matrix map,
the Cholesky/congruence, the per-pair eigenvalues and exp(-beta d^2) all happen inside one HIP launch.
"""
import torch

from .. import _lib, ops
from .._compat import GreaterThan, Kernel


class _BetaKernel(Kernel):
    """raw_beta parameter + GreaterThan(beta_min) constraint + beta property (kernels_spd.py:33-70)."""

    def __init__(self, beta_min, beta_prior=None, **kwargs):
        super().__init__(has_lengthscale=False, **kwargs)
        self.beta_min = beta_min
        self.register_parameter(name="raw_beta", parameter=torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, 1)))
        if beta_prior is not None:
            self.register_prior("beta_prior", beta_prior, lambda: self.beta, lambda v: self._set_beta(v))
        self.register_constraint("raw_beta", GreaterThan(self.beta_min))

    @property
    def beta(self):
        return self.raw_beta_constraint.transform(self.raw_beta)

    @beta.setter
    def beta(self, value):
        self._set_beta(value)

    def _set_beta(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_beta)
        self.initialize(raw_beta=self.raw_beta_constraint.inverse_transform(value))


def _diag_ones(x2):
    # diagonal_distance=True: the reference returns exp(-beta * zeros(..., N2, 1)) (spd_utils_torch.py:72-75)
    return torch.ones(tuple(x2.shape[:-1]) + (1,), dtype=torch.float64, device=x2.device)


class SpdAffineInvariantGaussianKernel(_BetaKernel):
    """k(X, Y) = exp(-beta d_AI(X, Y)^2)   (kernels_spd.py:17-100)."""

    def forward(self, x1, x2, diagonal_distance=False, **params):
        if diagonal_distance is True:
            return _diag_ones(x2)
        return ops.spd_ai_kernel(x1, x2, self.beta.double(), _lib.GABO_OUT_GAUSSIAN)


class SpdAffineInvariantLaplaceKernel(_BetaKernel):
    """k(X, Y) = exp(-beta d_AI(X, Y))   (kernels_spd.py:103-187)."""

    def forward(self, x1, x2, diagonal_distance=False, **params):
        if diagonal_distance is True:
            return _diag_ones(x2)
        return ops.spd_ai_kernel(x1, x2, self.beta.double(), _lib.GABO_OUT_LAPLACE)
