"""fit_gpytorch_manifold: fit the surrogate's hyper-parameters on the product of the manifolds they live on
(BoManifolds/manifold_optimization/manifold_gp_fit.py:54-222): Euclidean for the usual raw parameters, a sphere per nested-sphere
axis, the Grassmannian for the nested-SPD projection matrix (a parameter `p` declares its manifold through the attribute
`<p>_manifold` of its module, exactly as in the reference: manifold_gp_fit.py:163-171).

The loss is -(log marginal likelihood + log priors)/n of gabotorch_amd.models.SingleTaskGP; every evaluation builds the Gram
matrix and its gradient with respect to all hyper-parameters (beta / lengthscale, outputscale, noise, axes, projection matrix)
through the HIP kernels.  The optimiser itself moves a few dozen numbers and is host-side numpy, as in the reference."""
import time
from operator import attrgetter

import numpy as np
import torch

from .conjugate_gradient import ConjugateGradient
from .host_manifolds import Euclidean, Product


class _MllProblem:
    def __init__(self, model, names, params, manifold):
        self.model, self.names, self.params, self.manifold = model, names, params, manifold
        self.n_evals = 0

    def _set(self, x):
        with torch.no_grad():
            for p, v in zip(self.params, x):
                p.copy_(torch.as_tensor(np.asarray(v), dtype=p.dtype).reshape(p.shape))
        self.model.invalidate()

    def cost(self, x):
        self._set(x)
        self.n_evals += 1
        with torch.no_grad():
            try:
                return float(-self.model.marginal_log_likelihood())
            except RuntimeError:            # K + noise I not positive definite at this trial point
                return float("inf")

    def egrad(self, x):
        self._set(x)
        for p in self.params:
            p.grad = None
        loss = -self.model.marginal_log_likelihood()
        loss.backward()
        out = []
        for p, v in zip(self.params, x):
            g = torch.zeros_like(p) if p.grad is None else p.grad
            out.append(g.detach().cpu().double().numpy().reshape(np.shape(v)))
        return out

    def grad(self, x):
        return self.manifold.egrad2rgrad(x, self.egrad(x))


def fit_gpytorch_manifold(model, solver=None, nb_init_candidates=200, last_x_as_candidate_prob=0.9, exclude=None,
                          keep_first_euclidean=True):
    """Fits `model` (gabotorch_amd.models.SingleTaskGP) in place; returns (model, info).

    As in the reference: the start point is the best of `nb_init_candidates` candidates - the current parameters (with probability
    `last_x_as_candidate_prob`) and random points of the product manifold, three quarters of which keep the current values of the
    leading Euclidean hyper-parameters (manifold_gp_fit.py:187-197) - then Riemannian conjugate gradients."""
    solver = solver or ConjugateGradient(maxiter=500)
    exclude = set(exclude or ())
    named = [(n, p) for n, p in model.named_parameters() if n not in exclude]
    for _, p in named:
        p.requires_grad_(True)
    names = [n for n, _ in named]
    params = [p for _, p in named]
    factors, x0 = [], []
    for n, p in named:
        try:
            man = attrgetter(n + "_manifold")(model)
            shape = man._shape
        except AttributeError:
            shape = (int(p.numel()),)
            man = Euclidean(*shape)
        factors.append(man)
        x0.append(p.detach().cpu().double().numpy().reshape(shape).copy())
    manifold = Product(factors)
    problem = _MllProblem(model, names, params, manifold)
    t1 = time.time()
    cands = [x0] if np.random.rand() < last_x_as_candidate_prob else []
    cands += [manifold.rand() for _ in range(nb_init_candidates - len(cands))]
    if keep_first_euclidean:
        eucl = [k for k, m in enumerate(factors) if isinstance(m, Euclidean)]
        for i in range(int(3 * nb_init_candidates / 4)):
            for k in eucl:
                cands[i][k] = x0[k].copy()
    costs = [problem.cost(c) for c in cands]
    x_init = cands[int(np.argmin(costs))]
    opt_x, log = solver.solve(problem, x=x_init)
    problem._set(opt_x)
    for p in params:
        p.grad = None
    info = {"fopt": problem.cost(opt_x), "wall_time": time.time() - t1, "opt_log": log, "init_cost": float(np.min(costs)),
            "cost_evals": problem.n_evals}
    model.invalidate()
    return model, info
