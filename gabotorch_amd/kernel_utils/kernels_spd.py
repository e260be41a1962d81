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

    def beta_float(self):
        """beta as a Python float (the fp64 value of the parameter, fp32 unless the module was cast), for the launch arguments of the fused paths: the bits of
        `float(self.beta.double())` from ONE tensor operation - softplus of the raw parameter; the constraint's lower bound is added as the
        float32 sum the transform forms (exactly rounded either way) - and remembered per raw value.  The property costs five small tensor
        operations, 25-30 us of a sweep's set-up."""
        raw = self.raw_beta
        con = self.raw_beta_constraint
        if raw.numel() != 1 or type(con) is not GreaterThan or getattr(con, "lower_bound", None) is None:
            return float(self.beta.double())
        key = (raw.detach().item(), id(con))
        held = self.__dict__.get("_beta_float_held")
        if held is None or held[0] != key:
            import numpy as np
            lb = con.__dict__.get("_lower_bound_float")
            if lb is None:
                lb = con.__dict__["_lower_bound_float"] = float(con.lower_bound)
            if raw.dtype not in (torch.float32, torch.float64) or con.lower_bound.dtype != torch.float32 and con.lower_bound.dtype != raw.dtype:
                return float(self.beta.double())
            with torch.no_grad():
                sp = torch.nn.functional.softplus(raw.detach()).item()          # (in the parameter's own precision)
            # the transform's `softplus(raw) + lower_bound.to(raw)`: one exactly rounded sum in the parameter's precision
            val = float(np.float32(sp) + np.float32(lb)) if raw.dtype == torch.float32 else sp + lb
            held = self.__dict__["_beta_float_held"] = (key, val)
        return held[1]

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


class SpdFrobeniusGaussianKernel(Kernel):
    """k(X, Y) = exp(-||X - Y + 1e-15||_F^2 / lengthscale^2) on Mandel inputs   (kernels_spd.py:190-241).
    Differentiable (first order) in x1, x2 and the lengthscale through the HIP backward."""

    def __init__(self, **kwargs):
        self.has_lengthscale = True
        super().__init__(has_lengthscale=True, ard_num_dims=None, **kwargs)

    def forward(self, x1, x2, diagonal_distance=False, **params):
        if diagonal_distance is True:
            return _diag_ones(x2)
        return ops.frobenius_kernel(x1, x2, _beta_from_lengthscale(self.lengthscale), _lib.GABO_OUT_GAUSSIAN)


class SpdLogEuclideanGaussianKernel(Kernel):
    """k(X, Y) = exp(-||logm X - logm Y + 1e-15||_F^2 / lengthscale^2)   (kernels_spd.py:244-313): O(N) matrix logarithms
    (one wave per matrix) + one pairwise Frobenius launch, instead of the reference's two Python loops."""

    def __init__(self, **kwargs):
        self.has_lengthscale = True
        super().__init__(has_lengthscale=True, ard_num_dims=None, **kwargs)

    def forward(self, x1, x2, diagonal_distance=False, **params):
        if diagonal_distance is True:
            return _diag_ones(x2)
        l1 = ops.spd_logm_mandel_diff(x1)
        l2 = l1 if x2 is x1 else ops.spd_logm_mandel_diff(x2)
        return ops.frobenius_kernel(l1, l2, _beta_from_lengthscale(self.lengthscale), _lib.GABO_OUT_GAUSSIAN)


def _beta_from_lengthscale(ls):
    # exp(-d^2 / l^2) = exp(-beta d^2) with beta = l^-2, kept in torch so the lengthscale stays differentiable for the GP fit
    ls = ls.double().reshape(())
    return 1.0 / (ls * ls)


class NestedSpdAffineInvariantGaussianKernel(_BetaKernel):
    """Affine-invariant Gaussian kernel after the nested projection Y = W^T X W, W in G(D, d)
    (kernel_utils/kernels_nested_spd.py:19-136).  The projection matrix is a plain parameter here (pymanopt's Grassmann
    manifold, used by the reference only to draw the initial W, is replaced by qr(randn))."""

    def __init__(self, dim, latent_dim, beta_min, beta_prior=None, **kwargs):
        super().__init__(beta_min, beta_prior, **kwargs)
        self.dim, self.latent_dim = dim, latent_dim
        q, _ = torch.linalg.qr(torch.randn(dim, latent_dim, dtype=torch.float64))
        self.register_parameter(name="raw_projection_matrix", parameter=torch.nn.Parameter(q.repeat(*self.batch_shape, 1, 1)))
        from ..manifold_optimization.host_manifolds import Grassmann
        self.raw_projection_matrix_manifold = Grassmann(dim, latent_dim)          # kernels_nested_spd.py:75,175

    @property
    def projection_matrix(self):
        return self.raw_projection_matrix

    @projection_matrix.setter
    def projection_matrix(self, value):
        self.initialize(raw_projection_matrix=value)

    def forward(self, x1, x2, diagonal_distance=False, **params):
        if diagonal_distance is True:
            return _diag_ones(x2)
        w = self.projection_matrix.double()
        beta = self.beta.double()
        if ops.nested_spd_gram_applicable(x1, x2, w, beta):      # nobody differentiates this evaluation: projection + factorisation in one launch
            return ops.nested_spd_gram(x1, x2, w, float(beta), _lib.GABO_METRIC_AFFINE_INVARIANT)
        p1 = ops.spd_project_diff(x1, w)
        p2 = p1 if x2 is x1 else ops.spd_project_diff(x2, w)
        return ops.spd_ai_kernel(p1, p2, beta, _lib.GABO_OUT_GAUSSIAN)


class NestedSpdLogEuclideanGaussianKernel(SpdLogEuclideanGaussianKernel):
    """Log-Euclidean Gaussian kernel after the nested projection   (kernels_nested_spd.py:139-246; the kernel examples/hd_gabo_spd.py:164 uses)."""

    def __init__(self, dim, latent_dim, **kwargs):
        super().__init__(**kwargs)
        self.dim, self.latent_dim = dim, latent_dim
        q, _ = torch.linalg.qr(torch.randn(dim, latent_dim, dtype=torch.float64))
        self.register_parameter(name="raw_projection_matrix", parameter=torch.nn.Parameter(q.repeat(*self.batch_shape, 1, 1)))
        from ..manifold_optimization.host_manifolds import Grassmann
        self.raw_projection_matrix_manifold = Grassmann(dim, latent_dim)          # kernels_nested_spd.py:75,175

    @property
    def projection_matrix(self):
        return self.raw_projection_matrix

    @projection_matrix.setter
    def projection_matrix(self, value):
        self.initialize(raw_projection_matrix=value)

    def forward(self, x1, x2, diagonal_distance=False, **params):
        if diagonal_distance is True:
            return _diag_ones(x2)
        w = self.projection_matrix.double()
        if ops.nested_spd_gram_applicable(x1, x2, w, self.lengthscale):      # projection + logm in one launch, then the Frobenius Gram
            return ops.nested_spd_gram(x1, x2, w, float(_beta_from_lengthscale(self.lengthscale)), _lib.GABO_METRIC_LOG_EUCLIDEAN)
        p1 = ops.spd_project_diff(x1, w)
        p2 = p1 if x2 is x1 else ops.spd_project_diff(x2, w)
        return super().forward(p1, p2)
