"""Reconstruction parameters of the nested-sphere mapping, with the reference's names and signatures
(BoManifolds/nested_mappings/nested_spheres_optimization.py:20-100): the distances to the axes that minimise the squared geodesic
error between the data and their reconstruction from the subsphere; unconstrained on a product of Euclidean lines through
r = pi * sigmoid(.)."""
import math

import numpy as np
import torch

from ..manifold_optimization.host_manifolds import Euclidean, Product
from ..Riemannian_utils.sphere_utils_torch import sphere_distance_torch
from .nested_spheres_utils import projection_from_subsphere_to_sphere


def min_error_reconstruction_cost(x_data, x_subsphere, sphere_axes, sphere_distances):
    """sum_n d(x_n, reconstruction(x_subsphere_n))^2   (nested_spheres_optimization.py:20-38)."""
    x_rec = projection_from_subsphere_to_sphere(x_subsphere, sphere_axes, sphere_distances)[-1]
    cost = sphere_distance_torch(x_data.to(x_rec.device), x_rec, diag=True)
    return torch.sum(cost * cost)


def optimize_reconstruction_parameters_nested_sphere(x_data, x_subsphere, sphere_axes, solver, nb_init_candidates=100):
    """-> list of distances to the axes [S^d, ..., S^(d-r+1)] (1 x 1 tensors)   (nested_spheres_optimization.py:41-100)."""
    dev, dt = x_data.device, torch.float64
    x_data, x_subsphere = x_data.to(dt), x_subsphere.to(dev, dt)
    axes = [a.detach().to(dev, dt) for a in sphere_axes]
    n_levels = x_data.shape[1] - x_subsphere.shape[1]
    manifold = Product([Euclidean(1) for _ in range(n_levels)])

    def radii(params):
        return [math.pi * torch.sigmoid(p).reshape(1, 1) for p in params]          # gpytorch Interval(0, pi).transform

    def cost_torch(params):
        return min_error_reconstruction_cost(x_data, x_subsphere, axes, radii(params))

    def value_and_egrad(x):
        p = [torch.tensor(np.asarray(xi), dtype=dt, device=dev, requires_grad=True) for xi in x]
        v = cost_torch(p)
        grads = torch.autograd.grad(v, p, allow_unused=True)
        return float(v.detach()), [np.zeros(1) if g is None else g.detach().cpu().numpy().reshape(1) for g in grads]

    class _Problem:
        pass
    problem = _Problem()
    problem.manifold = manifold

    def value_only(x):                 # line searches and the candidate screening need no autograd graph
        with torch.no_grad():
            return float(cost_torch([torch.tensor(np.asarray(xi), dtype=dt, device=dev) for xi in x]))
    problem.cost = value_only
    problem.grad = lambda x: manifold.egrad2rgrad(x, value_and_egrad(x)[1])
    cands = [manifold.rand() for _ in range(nb_init_candidates)]
    vals = [problem.cost(c) for c in cands]
    opt, log = solver.solve(problem, x=cands[int(np.argmin(vals))])
    optimize_reconstruction_parameters_nested_sphere.last_log = dict(log, init_cost=float(np.min(vals)))
    return [r.detach() for r in radii([torch.tensor(o, dtype=dt, device=dev) for o in opt])]
