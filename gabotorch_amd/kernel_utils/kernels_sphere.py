"""GPyTorch kernels on the sphere S^n, computed by libgabo_hip.so on the MI355X.
Same class names and signatures as the reference (BoManifolds/kernel_utils/kernels_sphere.py:15-134)."""
from .. import _lib, ops
from .._compat import Kernel
from .kernels_spd import _BetaKernel


class SphereGaussianKernel(_BetaKernel):
    """k(x, y) = exp(-beta acos(<x, y>)^2)   (kernels_sphere.py:15-94)."""

    def forward(self, x1, x2, diag=False, **params):
        return ops.sphere_kernel(x1, x2, self.beta.double(), _lib.GABO_OUT_GAUSSIAN, diag=diag)


class SphereLaplaceKernel(Kernel):
    """k(x, y) = exp(-acos(<x, y>) / lengthscale^2), gpytorch `lengthscale` parameter   (kernels_sphere.py:97-134)."""

    def __init__(self, **kwargs):
        self.has_lengthscale = True
        super().__init__(has_lengthscale=True, ard_num_dims=None, **kwargs)

    def forward(self, x1, x2, diag=False, **params):
        ls = self.lengthscale.double()
        return ops.sphere_kernel(x1, x2, 1.0 / (ls * ls), _lib.GABO_OUT_LAPLACE, diag=diag)
