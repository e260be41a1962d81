"""Reconstruction parameters of the nested-sphere mapping, with the reference's names and signatures
(BoManifolds/nested_mappings/nested_spheres_optimization.py:20-100): the distances to the axes that minimise the squared geodesic
error between the data and their reconstruction from the subsphere; unconstrained on a product of Euclidean lines through
r = pi * sigmoid(.)."""
import math

import numpy as np
import torch

from ..manifold_optimization.host_manifolds import Euclidean
from ..Riemannian_utils.sphere_utils_torch import sphere_distance_torch
from .nested_spheres_utils import projection_from_subsphere_to_sphere


def min_error_reconstruction_cost(x_data, x_subsphere, sphere_axes, sphere_distances):
    """sum_n d(x_n, reconstruction(x_subsphere_n))^2   (nested_spheres_optimization.py:20-38)."""
    x_rec = projection_from_subsphere_to_sphere(x_subsphere, sphere_axes, sphere_distances)[-1]
    cost = sphere_distance_torch(x_data.to(x_rec.device), x_rec, diag=True)
    return torch.sum(cost * cost)


def optimize_reconstruction_parameters_nested_sphere(x_data, x_subsphere, sphere_axes, solver, nb_init_candidates=100):
    """-> list of distances to the axes [S^d, ..., S^(d-r+1)] (1 x 1 tensors)   (nested_spheres_optimization.py:41-100).
    One launch per evaluation (gabo_nested_sphere_reconstruction: every level of the lift, the distance to the data and the gradient with
    respect to the distances; the start candidates are ONE launch); the chain through r = pi * sigmoid(.) is host arithmetic."""
    from .. import ops
    dev, dt = x_data.device, torch.float64
    rec = ops.NestedSphereReconstruction(x_data.to(dt), x_subsphere.to(dev, dt), [a.detach() for a in sphere_axes])
    n_levels = rec.L
    # the reference's product of n_levels Euclidean lines (:62-63) IS R^n_levels: one vector, no Python loop over the factors per operation
    manifold = Euclidean(n_levels)

    def sigmoid(x):
        return 0.5 * (1.0 + np.tanh(0.5 * np.asarray(x, dtype=np.float64).reshape(-1)))      # = 1 / (1 + exp(-x)), without the overflow of exp far out

    recent = []                        # (key, value, gradient): a line search asks for the value, the solver then for the gradient there

    def value_and_egrad(x):
        sg = sigmoid(x)                                                # gpytorch Interval(0, pi).transform
        key = sg.tobytes()
        for k, v, g in recent:
            if k == key:
                return v, g
        v, g = rec.evaluate(math.pi * sg, grad=True)
        out = (float(v), g * math.pi * sg * (1.0 - sg))
        recent[:] = [(key,) + out] + recent[:3]
        return out

    def prefetch(xs):
        """the points a line search asks for next (first trial step and its first contraction) in ONE launch: blocks of different
        parameter sets run side by side, so the pair costs what one evaluation costs"""
        sgs = [sigmoid(x) for x in xs]
        sgs = [sg for sg in sgs if all(sg.tobytes() != k for k, _, _ in recent)]
        if len(sgs) < 2:
            return
        vals, grads = rec.evaluate(math.pi * np.stack(sgs), grad=True)
        new = [(sg.tobytes(), float(v), g * math.pi * sg * (1.0 - sg)) for sg, v, g in zip(sgs, vals, grads)]
        recent[:] = new + recent[:4 - len(new)]

    class _Problem:
        pass
    problem = _Problem()
    problem.manifold = manifold
    if n_levels >= 8:                  # (measured: 16.5 -> 13.6 ms at 48 levels; at 2 - 18 levels the second parameter set costs what the saved launch gains)
        problem.prefetch = prefetch
    problem.cost = lambda x: value_and_egrad(x)[0]
    problem.grad = lambda x: manifold.egrad2rgrad(x, value_and_egrad(x)[1])
    cands = [manifold.rand() for _ in range(nb_init_candidates)]
    vals = rec.evaluate(math.pi * np.stack([sigmoid(c) for c in cands]), grad=False)          # the candidate screening: one launch
    opt, log = solver.solve(problem, x=cands[int(np.argmin(vals))])
    optimize_reconstruction_parameters_nested_sphere.last_log = dict(log, init_cost=float(np.min(vals)))
    return [torch.tensor([[ri]], dtype=dt, device=dev) for ri in math.pi * sigmoid(opt)]
