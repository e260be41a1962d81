"""NestedSphereGaussianKernel with the reference's constructor, parameters and forward signature
(BoManifolds/kernel_utils/kernels_nested_sphere.py:19-152): the inputs on S^(dim-1) are projected level by level onto the latent
sphere (nested_spheres_utils) and the Gaussian sphere kernel is evaluated there, all on the MI355X and differentiable in the
inputs, the axes and beta."""
import numpy as np
import torch

from .. import _lib, ops
from ..manifold_optimization.host_manifolds import Sphere as HostSphere
from ..nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere
from .kernels_spd import _BetaKernel


class NestedSphereGaussianKernel(_BetaKernel):
    def __init__(self, dim, latent_dim, beta_min, beta_prior=None, **kwargs):
        super().__init__(beta_min, beta_prior, **kwargs)
        self.dim, self.latent_dim = dim, latent_dim
        for d in range(self.dim, self.latent_dim, -1):
            axis = torch.randn(1, d)
            axis = axis / torch.norm(axis)
            self.register_parameter(name="raw_axis_S" + str(d), parameter=torch.nn.Parameter(axis.repeat(*self.batch_shape, 1, 1)))
            setattr(self, "raw_axis_S" + str(d) + "_manifold", HostSphere(d))        # kernels_nested_sphere.py:88-90
        # distance to each axis fixed at pi/2: great subspheres  (kernels_nested_sphere.py:93-94)
        self.distances_to_axis = [np.pi / 2 * torch.ones(1, 1) for _ in range(self.dim, self.latent_dim, -1)]

    @property
    def axes(self):
        return [self._parameters["raw_axis_S" + str(d)] for d in range(self.dim, self.latent_dim, -1)]

    @axes.setter
    def axes(self, values_list):
        self._set_axes(values_list)

    def _set_axes(self, values_list):
        for d in range(self.dim, self.latent_dim, -1):
            value = values_list[self.dim - d]
            name = "raw_axis_S" + str(d)
            if not torch.is_tensor(value):
                value = torch.as_tensor(value)
            self.initialize(**{name: value.to(self._parameters[name])})

    def forward(self, x1, x2, diag=False, **params):
        axes = [a.double() for a in self.axes]
        px1 = projection_from_sphere_to_subsphere(x1, axes, self.distances_to_axis)[-1]
        px2 = px1 if x2 is x1 else projection_from_sphere_to_subsphere(x2, axes, self.distances_to_axis)[-1]
        return ops.sphere_kernel(px1, px2, self.beta.double(), _lib.GABO_OUT_GAUSSIAN, diag=diag)
