"""Reconstruction parameters of the nested-SPD mapping, with the reference's names and signatures
(BoManifolds/nested_mappings/nested_spd_optimization.py:23-186): the costs are sums of squared affine-invariant or log-Euclidean
distances between the data and their reconstructions, minimised over  V in G(D, D-d),  C in S^(D-d)_++,  K = t * unit vector reshaped,
t = sigmoid(.) in (0, 1), under W^T V = 0 by the augmented Lagrangian method.  Both built-in costs are ONE HIP launch per evaluation,
value and gradient together (gabo_nested_spd_reconstruction: csrc/nested_spd_reconstruction.hip)."""
import os
import threading
import weakref

import numpy as np
import torch

from .. import _lib, ops
from ..manifold_optimization.augmented_lagrange_method import AugmentedLagrangeMethod, _Constraint
from ..manifold_optimization.conjugate_gradient import ConjugateGradient
from ..manifold_optimization.host_manifolds import Euclidean, Grassmann, PositiveDefinite, Product, Sphere


def _default_threads():
    """0 = the native loop decides (spinning helper threads on hosts with >= 16 runnable cores); 1 when several ranks share the host: the
    spinning workers of every rank would compete for the same cores."""
    import torch.distributed as dist
    return "1" if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else "0"


def _composed_cost(metric, x_data, x_data_projected, projection_matrix, projection_complement_matrix, bottom_spd_matrix, contraction_matrix):
    """The same cost as a composition of this package's differentiable operations (lift, then N distances / matrix logarithms): gradients
    with respect to EVERY argument, the data, the latent points and W included, as the reference's torch statement gives them."""
    from ..Riemannian_utils.spd_utils_torch import affine_invariant_distance_torch, logm_torch
    from .nested_spd_utils import projection_from_nested_spd_to_spd
    x_rec = projection_from_nested_spd_to_spd(x_data_projected, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    if metric == _lib.GABO_RECON_AFFINE_INVARIANT:
        dist = affine_invariant_distance_torch(x_data.to(x_rec.device)[:, None], x_rec[:, None])        # batch N of 1 x 1 problems
        return torch.sum(dist * dist)
    diff = logm_torch(x_data.to(x_rec.device)) - logm_torch(x_rec) + 1e-15
    return torch.sum(diff * diff)


_prepared = {}      # (metric, data pointers / versions) -> NestedSpdReconstruction of the most recent data set
_prepared_lock = threading.Lock()


def clear_prepared_reconstruction():
    """Drops the cached preparation of the most recent data set (device workspaces included).  Call it after modifying the data tensors behind
    autograd's back (`.data`, memory shared with numpy): such edits do not move `_version`, which is what the cache key watches."""
    with _prepared_lock:
        _prepared.clear()


def _fused_cost(metric, x_data, x_data_projected, projection_matrix, projection_complement_matrix, bottom_spd_matrix, contraction_matrix):
    # the fused launch propagates gradients to V, C and K only: with a gradient requested for the data side, the composed path
    if torch.is_grad_enabled() and any(torch.is_tensor(a) and a.requires_grad for a in (x_data, x_data_projected, projection_matrix)):
        return _composed_cost(metric, x_data, x_data_projected, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                              contraction_matrix)
    # the prepared object (logm / inverse factors of the data, square roots of the latent points: two launches and a status read-back) is
    # kept for as long as the caller comes back with the same, unmodified tensors AND keeps them alive: the entry holds them weakly, so it
    # dies with the data set instead of pinning it (and its device workspaces) for the life of the process
    tensors = (x_data, x_data_projected, projection_matrix)
    key = (int(metric),) + tuple((a.data_ptr(), a._version, tuple(a.shape), str(a.device)) for a in tensors)
    with _prepared_lock:
        rec = _prepared.get("rec") if _prepared.get("key") == key else None
        if rec is not None and any(r() is not a for r, a in zip(_prepared["refs"], tensors)):
            rec = None      # same addresses, other tensor objects: the memory was recycled
        if rec is None:
            rec = ops.NestedSpdReconstruction(x_data, x_data_projected, projection_matrix, metric)
            _prepared["key"], _prepared["rec"] = key, rec
            _prepared["refs"] = tuple(weakref.ref(a, lambda _r: _prepared.clear()) for a in tensors)
    return rec(projection_complement_matrix, bottom_spd_matrix, contraction_matrix)


def min_affine_invariant_distance_reconstruction_cost(x_data, x_data_projected, projection_matrix, projection_complement_matrix,
                                                      bottom_spd_matrix, contraction_matrix):
    """sum_n d_AI(X_n, reconstruction(Y_n))^2   (nested_spd_optimization.py:23-56); differentiable in V, C, K."""
    return _fused_cost(_lib.GABO_RECON_AFFINE_INVARIANT, x_data, x_data_projected, projection_matrix, projection_complement_matrix,
                       bottom_spd_matrix, contraction_matrix)


def min_log_euclidean_distance_reconstruction_cost(x_data, x_data_projected, projection_matrix, projection_complement_matrix,
                                                   bottom_spd_matrix, contraction_matrix):
    """sum_n ||logm X_n - logm reconstruction(Y_n) + 1e-15||_F^2   (nested_spd_optimization.py:59-92); differentiable in V, C, K."""
    return _fused_cost(_lib.GABO_RECON_LOG_EUCLIDEAN, x_data, x_data_projected, projection_matrix, projection_complement_matrix,
                       bottom_spd_matrix, contraction_matrix)


def optimize_reconstruction_parameters_nested_spd(x_data, x_data_projected, projection_matrix, inner_solver,
                                                  cost_function=min_affine_invariant_distance_reconstruction_cost,
                                                  nb_init_candidates=100, maxiter=50, hip_graphs=True, native=True, alm_options=None):
    """-> (projection_complement_matrix D x (D-d), bottom_spd_matrix (D-d) x (D-d), contraction_matrix d x (D-d))
    (nested_spd_optimization.py:95-186).  The two built-in costs are served by the fused launch (the parameters stay host-resident numpy
    arrays as in the reference; an evaluation is one pinned copy in, one launch, one copy out; the nb_init_candidates start points are ONE
    launch); any other cost_function is differentiated by autograd, evaluation by evaluation.  hip_graphs: kept for signature compatibility.
    native: with a built-in cost and this package's ConjugateGradient as the inner solver (what the HD-GaBO example passes), the whole
    augmented-Lagrangian run is one call of the native host loop (gabo_nested_spd_reconstruction_solve: the same algorithm in C++ around
    the same launch - the numpy bookkeeping between two launches cost as much as the launches); False keeps the Python loop.
    alm_options: further keyword arguments of AugmentedLagrangeMethod (the reference fixes them: lambdas_fact = 0.05 and the defaults)."""
    dev, dt = x_data.device, torch.float64
    x_data, x_data_projected, W = x_data.to(dt), x_data_projected.to(dev, dt), projection_matrix.to(dev, dt)
    dim, latent = x_data.shape[1], W.shape[1]
    comp = dim - latent
    manifold = Product([Grassmann(dim, comp), PositiveDefinite(comp), Sphere(latent * comp), Euclidean(1)])
    shapes = [(dim, comp), (comp, comp), (latent * comp,), (1,)]
    sizes = [int(np.prod(sh)) for sh in shapes]
    W_host = W.cpu().numpy()

    class _OrthogonalityConstraint:
        """||V^T W||_F = 0 (W^T V = 0, :142-147) and its Euclidean gradient W (W^T V) / ||V^T W||_F with respect to V, in numpy: a
        few hundred flops on a host-resident parameter."""

        @staticmethod
        def cost(x):
            return float(np.linalg.norm(np.asarray(x[0]).T @ W_host))

        def __call__(self, x):
            V = np.asarray(x[0], dtype=np.float64)
            wtv = W_host.T @ V
            value = float(np.linalg.norm(wtv))
            gV = (W_host @ wtv) / value if value > 0.0 else np.zeros_like(V)
            return value, [gV] + [np.zeros(np.shape(xi)) for xi in x[1:]]

    def contraction(x):
        """K = sigmoid(x[3]) * x[2] reshaped (gpytorch Interval(0, 1).transform, :139, 155) -> (K, t, unit matrix)"""
        t = 1.0 / (1.0 + np.exp(-float(np.asarray(x[3]).reshape(-1)[0])))
        unit = np.asarray(x[2], dtype=np.float64).reshape(latent, comp)
        return t * unit, t, unit

    class _FusedEvaluator:
        """value / (value, Euclidean gradient) at a point through gabo_nested_spd_reconstruction.  Every evaluation is the value AND the
        gradient (the launch is two dependent eigen-solves either way; the adjoint adds ~15 % to it): the line search asks for values, and
        the solver then asks for the gradient at the point the search accepted - the last or second-to-last one it evaluated - so the two
        most recent points are remembered and that request costs nothing (a third fewer launches per augmented-Lagrangian run)."""

        def __init__(self, metric):
            self.rec = ops.NestedSpdReconstruction(x_data, x_data_projected, W, metric)
            self.recent = []                 # [(key, value, grads)], newest first

        def many(self, xs):
            ks = [contraction(x)[0] for x in xs]
            return self.rec.evaluate_host(np.stack([x[0] for x in xs]), np.stack([x[1] for x in xs]), np.stack(ks), grad=False)

        @staticmethod
        def _key(x):
            return b"".join(np.ascontiguousarray(p, dtype=np.float64).tobytes() for p in x)

        @staticmethod
        def _chain(x, t, unit, gV, gC, gK):
            return [gV, gC, (t * gK).reshape(np.shape(x[2])), np.full(np.shape(x[3]), float(np.sum(gK * unit)) * t * (1.0 - t))]

        def __call__(self, x):
            key = self._key(x)
            for k, value, grads in self.recent:
                if k == key:
                    return value, grads
            K, t, unit = contraction(x)
            value, gV, gC, gK = self.rec.evaluate_host(x[0], x[1], K, grad=True)
            grads = self._chain(x, t, unit, gV, gC, gK)
            self.recent = [(key, float(value), grads)] + self.recent[:3]
            return float(value), grads

        def prefetch(self, xs):
            """the points a line search will ask for next, in ONE launch (blocks of different parameter sets run side by side: the
            launch takes as long as a single evaluation)"""
            xs = [x for x in xs if all(self._key(x) != k for k, _, _ in self.recent)]
            if len(xs) < 2:
                return
            parts = [contraction(x) for x in xs]
            values, gV, gC, gK = self.rec.evaluate_host(np.stack([x[0] for x in xs]), np.stack([x[1] for x in xs]),
                                                        np.stack([pt[0] for pt in parts]), grad=True)
            new = [(self._key(x), float(values[i]), self._chain(x, parts[i][1], parts[i][2], gV[i], gC[i], gK[i])) for i, x in enumerate(xs)]
            self.recent = new + self.recent[:4 - len(new)]

        def cost(self, x):
            return self(x)[0]

    class _AutogradEvaluator:
        """the same for a user-supplied cost_function(x_data, x_data_projected, W, V, C, K) -> 0-dim tensor: torch autograd, eager"""

        def __init__(self):
            self.key, self.value, self.grads = None, None, None

        def _at(self, x):
            key = b"".join(np.ascontiguousarray(p, dtype=np.float64).tobytes() for p in x)
            if key != self.key:
                self.key, self.value, self.grads = key, None, None

        def _eval(self, x, with_grad):
            parts = [torch.tensor(np.asarray(p, dtype=np.float64).reshape(sh), device=dev, requires_grad=with_grad) for p, sh in zip(x, shapes)]
            with torch.set_grad_enabled(with_grad):
                K = torch.sigmoid(parts[3]) * parts[2].reshape(latent, comp)
                v = cost_function(x_data, x_data_projected, W, parts[0], parts[1], K)
            if not with_grad:
                return float(v), None
            grads = torch.autograd.grad(v, parts, allow_unused=True)
            return float(v), [np.zeros(sh) if g is None else g.cpu().numpy().reshape(np.shape(xi)) for g, sh, xi in zip(grads, shapes, x)]

        def many(self, xs):
            return np.array([self._eval(x, False)[0] for x in xs])

        def cost(self, x):
            self._at(x)
            if self.value is None:
                self.value = self._eval(x, False)[0]
            return self.value

        def __call__(self, x):
            self._at(x)
            if self.grads is None:
                self.value, self.grads = self._eval(x, True)
            return self.value, self.grads

    if cost_function is min_log_euclidean_distance_reconstruction_cost:
        cost_vg = _FusedEvaluator(_lib.GABO_RECON_LOG_EUCLIDEAN)
    elif cost_function is min_affine_invariant_distance_reconstruction_cost:
        cost_vg = _FusedEvaluator(_lib.GABO_RECON_AFFINE_INVARIANT)
    else:
        cost_vg = _AutogradEvaluator()

    class _Problem:
        pass
    problem = _Problem()
    problem.manifold = manifold
    problem.cost = cost_vg.cost
    problem.grad = lambda x: manifold.egrad2rgrad(x, cost_vg(x)[1])
    if hasattr(cost_vg, "prefetch"):
        problem.prefetch = cost_vg.prefetch
    con_vg = _OrthogonalityConstraint()
    constraint = _Constraint(manifold, con_vg)
    constraint.cost = con_vg.cost
    cands = [manifold.rand() for _ in range(nb_init_candidates)]
    vals = cost_vg.many(cands)                                           # best of the random starts (:158-166): one launch
    x0 = cands[int(np.argmin(vals))]
    alm_options = dict(alm_options or {})
    logverbosity = int(alm_options.get("logverbosity", 0) or 0)
    solver = AugmentedLagrangeMethod(**dict(dict(maxiter=maxiter, inner_solver=inner_solver, lambdas_fact=0.05), **alm_options))
    # the native loop implements the algorithm with the options below; anything else in alm_options (multiplier update factor, logging, line
    # search settings of another inner solver ...) is served by the Python loop
    _NATIVE_KEYS = {"maxiter", "bound", "rho_init", "thetarho", "tau", "starting_tolgradnorm", "ending_tolgradnorm", "gammas_fact", "minstepsize",
                    "maxtime"}
    if native and isinstance(cost_vg, _FusedEvaluator) and type(inner_solver) is ConjugateGradient and set(alm_options) <= _NATIVE_KEYS:
        options = _lib.ReconSolveOptions(
            bound=solver._bound, rho_init=solver._rho_init, thetarho=solver._thetarho, tau=solver._tau,
            starting_tolgradnorm=solver._starting_tolgradnorm, ending_tolgradnorm=solver._ending_tolgradnorm, gammas_fact=solver._gammas_fact,
            minstepsize=solver._minstepsize, maxtime=solver._maxtime, maxiter=solver._maxiter, cg_minstepsize=inner_solver.minstepsize,
            cg_maxtime=inner_solver.maxtime, cg_orth_value=inner_solver.orth_value, cg_maxiter=inner_solver.maxiter,
            lookahead=int(os.environ.get("GABO_RECON_LOOKAHEAD", "0")), host_threads=int(os.environ.get("GABO_RECON_THREADS", _default_threads())))
        v, c, unit, raw, log = cost_vg.rec.solve_host(x0[0], x0[1], x0[2], x0[3], options)
        opt = [v, c, unit, raw]
        optimize_reconstruction_parameters_nested_spd.last_log = dict(log, init_cost=float(np.min(vals)), native=True)
        return (torch.tensor(v, dtype=dt, device=dev), torch.tensor(c, dtype=dt, device=dev),
                torch.tensor(contraction(opt)[0], dtype=dt, device=dev))
    opt = solver.solve(problem, x=x0, eq_constraints=[constraint])
    if logverbosity >= 1:                                                # pymanopt convention: (x, log) with logverbosity >= 1
        opt = opt[0]
    V = torch.tensor(opt[0], dtype=dt, device=dev)
    C = torch.tensor(opt[1], dtype=dt, device=dev)
    K = torch.tensor(contraction(opt)[0], dtype=dt, device=dev)
    optimize_reconstruction_parameters_nested_spd.last_log = dict(solver.log, init_cost=float(np.min(vals)), final_cost=problem.cost(opt))
    return V, C, K
