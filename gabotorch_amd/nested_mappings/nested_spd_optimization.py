"""Reconstruction parameters of the nested-SPD mapping, with the reference's names and signatures
(BoManifolds/nested_mappings/nested_spd_optimization.py:23-186): the costs are sums of squared affine-invariant or log-Euclidean
distances between the data and their reconstructions (HIP kernels, differentiable through the closed-form backward kernels and the
matrix-function adjoint), minimised over  V in G(D, D-d),  C in S^(D-d)_++,  K = t * unit vector reshaped, t = sigmoid(.) in (0, 1),
under W^T V = 0 by the augmented Lagrangian method."""
import numpy as np
import torch

from ..manifold_optimization.augmented_lagrange_method import AugmentedLagrangeMethod, _Constraint
from ..manifold_optimization.host_manifolds import Euclidean, Grassmann, PositiveDefinite, Product, Sphere
from ..Riemannian_utils.spd_utils_torch import affine_invariant_distance_torch, logm_torch
from .nested_spd_utils import projection_from_nested_spd_to_spd


def min_affine_invariant_distance_reconstruction_cost(x_data, x_data_projected, projection_matrix, projection_complement_matrix,
                                                      bottom_spd_matrix, contraction_matrix):
    """sum_n d_AI(X_n, reconstruction(Y_n))^2   (nested_spd_optimization.py:23-56).  One batched launch for the N distances."""
    x_rec = projection_from_nested_spd_to_spd(x_data_projected, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    dist = affine_invariant_distance_torch(x_data.to(x_rec.device)[:, None], x_rec[:, None])        # batch N of 1 x 1 problems
    return torch.sum(dist * dist)


def min_log_euclidean_distance_reconstruction_cost(x_data, x_data_projected, projection_matrix, projection_complement_matrix,
                                                   bottom_spd_matrix, contraction_matrix):
    """sum_n ||logm X_n - logm reconstruction(Y_n) + 1e-15||_F^2   (nested_spd_optimization.py:59-92)."""
    x_rec = projection_from_nested_spd_to_spd(x_data_projected, projection_matrix, projection_complement_matrix, bottom_spd_matrix,
                                              contraction_matrix)
    diff = logm_torch(x_data.to(x_rec.device)) - logm_torch(x_rec) + 1e-15
    return torch.sum(diff * diff)


def optimize_reconstruction_parameters_nested_spd(x_data, x_data_projected, projection_matrix, inner_solver,
                                                  cost_function=min_affine_invariant_distance_reconstruction_cost,
                                                  nb_init_candidates=100, maxiter=50, hip_graphs=True):
    """-> (projection_complement_matrix D x (D-d), bottom_spd_matrix (D-d) x (D-d), contraction_matrix d x (D-d))
    (nested_spd_optimization.py:95-186).  hip_graphs (extension): replay the evaluations of the two built-in costs from hipGraphs."""
    dev, dt = x_data.device, torch.float64
    x_data, x_data_projected, W = x_data.to(dt), x_data_projected.to(dev, dt), projection_matrix.to(dev, dt)
    dim, latent = x_data.shape[1], W.shape[1]
    manifold = Product([Grassmann(dim, dim - latent), PositiveDefinite(dim - latent), Sphere(latent * (dim - latent)), Euclidean(1)])

    shapes = [(dim, dim - latent), (dim - latent, dim - latent), (latent * (dim - latent),), (1,)]
    sizes = [int(np.prod(sh)) for sh in shapes]

    # what does not depend on the parameters is evaluated once: logm of the data (log-Euclidean cost) and sqrtm of the latent points
    from .. import _lib, ops
    sqrt_low = ops.spd_matrix_function(x_data_projected, _lib.GABO_SPD_SQRTM).to(dt)
    if cost_function is min_log_euclidean_distance_reconstruction_cost:
        log_data = logm_torch(x_data)

        def data_cost(V, C, K):
            x_rec = projection_from_nested_spd_to_spd(x_data_projected, W, V, C, K, sqrt_low=sqrt_low)
            diff = log_data - logm_torch(x_rec) + 1e-15
            return torch.sum(diff * diff)
    elif cost_function is min_affine_invariant_distance_reconstruction_cost:
        def data_cost(V, C, K):
            x_rec = projection_from_nested_spd_to_spd(x_data_projected, W, V, C, K, sqrt_low=sqrt_low)
            dist = affine_invariant_distance_torch(x_data[:, None], x_rec[:, None])
            return torch.sum(dist * dist)
    else:
        def data_cost(V, C, K):
            return cost_function(x_data, x_data_projected, W, V, C, K)

    def cost_torch(p):
        norm = torch.sigmoid(p[3])                                       # gpytorch Interval(0, 1).transform   (:139,155)
        K = norm * p[2].reshape(latent, dim - latent)
        return data_cost(p[0], p[1], K)

    W_host = W.cpu().numpy()

    class _OrthogonalityConstraint:
        """||V^T W||_F = 0 (W^T V = 0, :142-147) and its Euclidean gradient W (W^T V) / ||V^T W||_F with respect to V, in numpy: a
        few hundred flops on a host-resident parameter (a torch CPU call here costs milliseconds of thread-pool wake-up on a
        many-core host, a device launch a round trip)."""

        @staticmethod
        def cost(x):
            return float(np.linalg.norm(np.asarray(x[0]).T @ W_host))

        def __call__(self, x):
            V = np.asarray(x[0], dtype=np.float64)
            wtv = W_host.T @ V
            value = float(np.linalg.norm(wtv))
            gV = (W_host @ wtv) / value if value > 0.0 else np.zeros_like(V)
            return value, [gV] + [np.zeros(np.shape(xi)) for xi in x[1:]]

    class _Evaluator:
        """value / (value, Euclidean gradient) of fn at a point, remembering the last point: the augmented Lagrangian asks for the
        cost and the gradient of the same point separately, and line searches only need values (no autograd graph).
        graphs=True: an evaluation is a fixed sequence of ~40 small launches on fixed shapes (two eigen-solves, a dozen products,
        their adjoints), so both variants are captured once into hipGraphs (torch.cuda.CUDAGraph) on a static parameter buffer and
        replayed: one host-to-device copy, one graph launch and one read-back per evaluation instead of ~40 eager launches."""

        def __init__(self, fn, device, graphs=False):
            self.fn, self.device, self.key, self.value, self.grads = fn, device, None, None, None
            self.graphs = bool(graphs) and torch.device(device).type == "cuda"
            self._captured = {}

        def _at(self, x):
            key = b"".join(np.ascontiguousarray(p, dtype=np.float64).tobytes() for p in x)
            if key != self.key:
                self.key, self.value, self.grads = key, None, None

        def _eval(self, flat, with_grad):
            parts = [t.reshape(sh) for t, sh in zip(torch.split(flat, sizes), shapes)]
            if not with_grad:
                with torch.no_grad():
                    return self.fn(parts).reshape(1)
            parts = [t.requires_grad_(True) for t in parts]
            v = self.fn(parts)
            grads = torch.autograd.grad(v, parts, allow_unused=True)
            return torch.cat([v.detach().reshape(1)] + [torch.zeros_like(pi).reshape(-1) if g is None else g.reshape(-1)
                                                        for g, pi in zip(grads, parts)])

        def _run(self, x, with_grad):
            host = torch.from_numpy(np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1) for p in x]))
            if not self.graphs:
                return self._eval(host.to(self.device), with_grad).cpu().numpy()
            ent = self._captured.get(with_grad)
            if ent is None:
                previous = ops.set_error_checking(False)          # the status read-back would synchronise inside the capture
                try:
                    static_in = host.to(self.device)
                    side = torch.cuda.Stream(device=self.device)
                    side.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(side):
                        for _ in range(3):                          # warm-up outside capture (lazy caches, allocator)
                            self._eval(static_in, with_grad)
                    torch.cuda.current_stream(self.device).wait_stream(side)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        out = self._eval(static_in, with_grad)
                finally:
                    ops.set_error_checking(previous)
                ent = self._captured[with_grad] = (graph, static_in, out)
            graph, static_in, out = ent
            static_in.copy_(host)
            graph.replay()
            return out.cpu().numpy()

        def cost(self, x):
            self._at(x)
            if self.value is None:
                self.value = float(self._run(x, False)[0])
            return self.value

        def __call__(self, x):
            self._at(x)
            if self.grads is None:
                flat = self._run(x, True)
                self.value = float(flat[0])
                self.grads = [a.reshape(np.shape(xi)) for a, xi in zip(np.split(flat[1:], np.cumsum(sizes)[:-1]), x)]
            return self.value, self.grads

    class _Problem:
        pass
    problem = _Problem()
    problem.manifold = manifold
    builtin_cost = cost_function in (min_log_euclidean_distance_reconstruction_cost, min_affine_invariant_distance_reconstruction_cost)
    cost_vg = _Evaluator(cost_torch, dev, graphs=hip_graphs and builtin_cost)      # (a user cost may synchronise: eager)
    problem.cost = cost_vg.cost
    problem.grad = lambda x: manifold.egrad2rgrad(x, cost_vg(x)[1])
    con_vg = _OrthogonalityConstraint()
    constraint = _Constraint(manifold, con_vg)
    constraint.cost = con_vg.cost
    cands = [manifold.rand() for _ in range(nb_init_candidates)]
    vals = [cost_vg.cost(c) for c in cands]
    x0 = cands[int(np.argmin(vals))]
    solver = AugmentedLagrangeMethod(maxiter=maxiter, inner_solver=inner_solver, lambdas_fact=0.05)
    opt = solver.solve(problem, x=x0, eq_constraints=[constraint])
    V = torch.tensor(opt[0], dtype=dt, device=dev)
    C = torch.tensor(opt[1], dtype=dt, device=dev)
    K = torch.sigmoid(torch.tensor(opt[3], dtype=dt, device=dev)) * torch.tensor(opt[2], dtype=dt, device=dev).reshape(latent, dim - latent)
    optimize_reconstruction_parameters_nested_spd.last_log = dict(solver.log, init_cost=float(np.min(vals)), final_cost=problem.cost(opt))
    return V, C, K
