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
from .host_manifolds import Euclidean, PackedEuclideanSpheres, Product, Sphere


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


class _NestedSpdMllProblem(_MllProblem):
    """The same objective for ScaleKernel(NestedSpd{LogEuclidean, AffineInvariant}GaussianKernel) - the surrogate of HD-GaBO
    (examples/hd_bo_spd/benchmark_examples/hd_gabo_spd.py:163-205) - without an autograd graph: an evaluation is the chain
        project (gabo_spd_project) -> [logm (gabo_spd_logm_mandel)] -> Gaussian Gram of the latent points -> gabo_gp_mll_gram
    and, for the gradient, the kernels' own backward launches in reverse (W = alpha alpha^T - Ky^-1 from the likelihood launch is
    d ll / d Gram up to outputscale / 2), the projection's adjoint dW = 2 sum_n X_n W G_n, and one read-back; the chain rule through
    softplus and the Gamma priors of the scalar hyper-parameters is Python float arithmetic (SingleTaskGP._fast_scalar_objective).
    Round 2 differentiated the same launches through torch autograd: 0.3 ms per value, 1.0 ms per gradient at n = 10."""

    @staticmethod
    def build(model, names, params, manifold):
        from .. import _compat, ops
        from ..kernel_utils import kernels_spd as kspd
        cm = model.covar_module
        base = getattr(cm, "base_kernel", None)
        if _compat.HAVE_GPYTORCH or base is None or type(base) not in (kspd.NestedSpdLogEuclideanGaussianKernel, kspd.NestedSpdAffineInvariantGaussianKernel):
            return None
        iw = [k for k, p in enumerate(params) if p is base.raw_projection_matrix]
        if len(iw) != 1 or base.raw_projection_matrix.dim() != 2 or not model.train_x.is_cuda and not torch.cuda.is_available():
            return None
        from .. import _lib
        # beyond what the one-call evaluators take (tiled likelihood: n <= 2048; projection kernels: D <= 32) the generic autograd problem is used
        if model.train_x.shape[0] > _lib.GABO_GP_MLL_LARGE_MAX_N or base.raw_projection_matrix.shape[0] > _lib.GABO_SPD_MAX_DIM:
            return None
        self = _NestedSpdMllProblem(model, names, params, manifold)
        self.iw = iw[0]
        self.scalar_idx = [k for k in range(len(params)) if k != self.iw]
        if any(params[k].numel() != 1 for k in self.scalar_idx):
            return None
        self.log_euclidean = type(base) is kspd.NestedSpdLogEuclideanGaussianKernel
        self.ops = ops
        dev = ops._device_for(model.train_x)
        self.x = model.train_x.to(dev).double().contiguous()
        self.xm = ops.mandel_to_matrix(self.x)                          # n x D x D, fixed during the fit (projection adjoint)
        self.y = model.train_y.to(dev).double().contiguous()
        self.W = None
        self.want_grad = False
        self.grad_w = None
        self.native, self.values_only, self.W_host, self._native_buffers, self._recent = True, False, None, None, []
        self.scalar = model._fast_scalar_objective([params[k] for k in self.scalar_idx], evaluator=self._evaluate)
        return self if self.scalar is not None else None

    def _native(self, theta, outputscale, noise, mean):
        """The log-Euclidean kernel's chain as ONE host call (gabo_nested_spd_fit_evaluate: the same launches issued from C++, one pinned
        copy in and one out) - always with the gradient: a line search asks for the value at a point and the solver for the gradient at the
        point it accepted, so the last results are remembered and that second request costs nothing."""
        import ctypes
        from .. import _lib
        lib = _lib.load()
        n, D, d = self.x.shape[0], self.W_host.shape[0], self.W_host.shape[1]
        if self._native_buffers is None:
            ws = torch.empty(max(int(lib.gabo_nested_spd_fit_workspace_bytes(n, D, d)), 16), dtype=torch.uint8, device=self.x.device)
            self._native_buffers = (ws, torch.empty(2 * D * d + 7, dtype=torch.float64).pin_memory(), np.empty(7 + D * d))
        ws, pinned, out = self._native_buffers
        key = (self.W_host.tobytes(), theta, outputscale, noise, mean)
        for k, res in self._recent:
            if k == key and (res[2] or not self.want_grad):
                self.grad_w = res[1]
                return res[0]
        want = self.want_grad or not self.values_only
        with torch.cuda.device(self.x.device):
            _lib.check(lib.gabo_nested_spd_fit_evaluate(
                self.x.data_ptr(), self.xm.data_ptr(), self.y.data_ptr(), self.W_host.ctypes.data_as(ctypes.c_void_p), n, D, d, float(theta),
                float(outputscale), float(noise), float(mean), 1 if want else 0, out.ctypes.data_as(ctypes.c_void_p), ws.data_ptr(), ws.numel(),
                pinned.data_ptr(), pinned.numel(), self.ops._stream_ptr(self.x.device)), "gabo_nested_spd_fit_evaluate")
        res = ((float(out[0]), float(out[6]), float(out[2]), float(out[3]), float(out[4]), float(out[5])), out[7:].reshape(D, d).copy(), want)
        self._recent = [(key, res)] + self._recent[:3]
        self.grad_w = res[1]
        return res[0]

    def _evaluate(self, theta, outputscale, noise, mean):
        from .. import _lib
        ops = self.ops
        if self.log_euclidean and self.native:
            return self._native(theta, outputscale, noise, mean)
        z = ops.spd_project(self.x, self.W)
        if self.log_euclidean:
            feat = ops.spd_logm_mandel(z)
            kb = ops.frobenius_pairwise(feat, feat, theta, _lib.GABO_OUT_GAUSSIAN)
        else:
            kb = ops.spd_ai_pairwise(z, z, theta, _lib.GABO_OUT_GAUSSIAN)
        out, wm = ops.gp_mll_gram(kb, self.y, outputscale, noise, mean, want_w=self.want_grad)
        if not self.want_grad:
            o = out.tolist()
            return o[0], 0.0, o[2], o[3], o[4], o[5]
        gkb = (0.5 * outputscale) * wm                                   # d ll / d kb
        g_theta = (gkb * torch.xlogy(kb, kb)).sum() / theta              # kb = exp(-theta E):  d kb / d theta = -E kb = kb log(kb) / theta
        if self.log_euclidean:
            gfeat = ops.frobenius_backward(feat, feat, gkb, theta, _lib.GABO_OUT_GAUSSIAN, wrt=1) \
                + ops.frobenius_backward(feat, feat, gkb, theta, _lib.GABO_OUT_GAUSSIAN, wrt=2)
            gz = ops.spd_logm_mandel_backward(z, gfeat)
        else:
            gz = ops.spd_ai_backward(z, z, gkb, theta, _lib.GABO_OUT_GAUSSIAN, wrt=1) + ops.spd_ai_backward(z, z, gkb, theta, _lib.GABO_OUT_GAUSSIAN, wrt=2)
        gm = ops.mandel_to_matrix(gz)                                    # n x d x d (symmetric)
        gw = 2.0 * torch.einsum("nab,bc,ncd->ad", self.xm, self.W, gm)
        flat = torch.cat([out, g_theta.reshape(1), gw.reshape(-1)]).tolist()          # one read-back
        self.grad_w = np.asarray(flat[7:]).reshape(self.W.shape)
        return flat[0], flat[6], flat[2], flat[3], flat[4], flat[5]

    def _run(self, x, want_grad):
        if self.log_euclidean and self.native:
            self.W_host = np.ascontiguousarray(np.asarray(x[self.iw], dtype=np.float64).reshape(self.params[self.iw].shape))
        else:
            self.W = torch.as_tensor(np.asarray(x[self.iw], dtype=np.float64), device=self.x.device).reshape(self.params[self.iw].shape).contiguous()
        self.want_grad = want_grad
        v = np.array([float(np.asarray(x[k]).reshape(-1)[0]) for k in self.scalar_idx])
        return self.scalar(v)

    def cost(self, x):
        self.n_evals += 1
        loss, _ = self._run(x, False)
        return float("inf") if loss >= 1e10 else float(loss)

    def egrad(self, x):
        loss, g = self._run(x, True)
        n = self.y.numel()
        out = [None] * len(self.params)
        for pos, k in enumerate(self.scalar_idx):
            out[k] = np.full(np.shape(x[k]), g[pos])
        out[self.iw] = (np.zeros(np.shape(x[self.iw])) if loss >= 1e10 else -self.grad_w.reshape(np.shape(x[self.iw])) / n)
        return out


class _NestedSphereMllProblem(_MllProblem):
    """The same objective for ScaleKernel(NestedSphereGaussianKernel) - the surrogate of HD-GaBO on the sphere
    (examples/hd_bo_sphere/benchmark_examples/hd_gabo_sphere.py:150-180): the axes of the nested spheres are learnt on their spheres together
    with the scalar hyper-parameters.  One evaluation is ONE host call (gabo_nested_sphere_fit_evaluate: every level of the projection for
    every training point in one launch, the Gram matrix, the likelihood, and their adjoints back to the axes); torch autograd through the
    level-by-level Python statement of the projection needed ~50 launches per level."""

    @staticmethod
    def build(model, names, params, manifold):
        from .. import _compat, ops
        from ..kernel_utils.kernels_nested_sphere import NestedSphereGaussianKernel
        cm = model.covar_module
        base = getattr(cm, "base_kernel", None)
        if _compat.HAVE_GPYTORCH or base is None or type(base) is not NestedSphereGaussianKernel:
            return None
        if not model.train_x.is_cuda and not torch.cuda.is_available():
            return None
        from .. import _lib
        if model.train_x.shape[0] > _lib.GABO_GP_MLL_LARGE_MAX_N:       # beyond the tiled likelihood: the generic autograd problem
            return None
        axis_idx = []
        for a in base.axes:
            hits = [k for k, p in enumerate(params) if p is a]
            if len(hits) != 1:
                return None
            axis_idx.append(hits[0])
        self = _NestedSphereMllProblem(model, names, params, manifold)
        self.axis_idx = axis_idx
        self.scalar_idx = [k for k in range(len(params)) if k not in axis_idx]
        if any(params[k].numel() != 1 for k in self.scalar_idx) or base.dim - base.latent_dim != len(axis_idx) or base.latent_dim > 64:
            return None
        self.ops = ops
        dev = ops._device_for(model.train_x)
        self.x = model.train_x.to(dev).double().contiguous()
        self.y = model.train_y.to(dev).double().contiguous()
        if self.x.dim() != 2 or self.x.shape[1] != base.dim:
            return None
        self.D, self.L = int(base.dim), len(axis_idx)
        self.dists = np.array([float(d.reshape(-1)[0]) for d in base.distances_to_axis], dtype=np.float64)
        self.total = sum(self.D - k for k in range(self.L))
        self.want_grad, self.values_only, self.axes_host, self.grad_axes, self._buffers, self._recent = False, False, None, None, None, []
        self.scalar = model._fast_scalar_objective([params[k] for k in self.scalar_idx], evaluator=self._evaluate)
        return self if self.scalar is not None else None

    def _evaluate(self, theta, outputscale, noise, mean):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        n = self.x.shape[0]
        if self._buffers is None:
            ws = torch.empty(max(int(lib.gabo_nested_sphere_fit_workspace_bytes(n, self.D, self.L)), 16), dtype=torch.uint8, device=self.x.device)
            self._buffers = (ws, torch.empty(2 * self.total + self.L + 7, dtype=torch.float64).pin_memory(), np.empty(7 + self.total))
        ws, pinned, out = self._buffers
        key = (self.axes_host.tobytes(), theta, outputscale, noise, mean)
        for k, res in self._recent:
            if k == key and (res[2] or not self.want_grad):
                self.grad_axes = res[1]
                return res[0]
        want = self.want_grad or not self.values_only
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        with torch.cuda.device(self.x.device):
            _lib.check(lib.gabo_nested_sphere_fit_evaluate(
                self.x.data_ptr(), self.y.data_ptr(), ptr(self.axes_host), ptr(self.dists), n, self.D, self.L, float(theta), float(outputscale),
                float(noise), float(mean), 1 if want else 0, ptr(out), ws.data_ptr(), ws.numel(), pinned.data_ptr(), pinned.numel(),
                self.ops._stream_ptr(self.x.device)), "gabo_nested_sphere_fit_evaluate")
        res = ((float(out[0]), float(out[6]), float(out[2]), float(out[3]), float(out[4]), float(out[5])), out[7:].copy(), want)
        self._recent = [(key, res)] + self._recent[:3]
        self.grad_axes = res[1]
        return res[0]

    def _run(self, x, want_grad):
        self.axes_host = np.ascontiguousarray(np.concatenate([np.asarray(x[k], dtype=np.float64).reshape(-1) for k in self.axis_idx]))
        self.want_grad = want_grad
        v = np.array([float(np.asarray(x[k]).reshape(-1)[0]) for k in self.scalar_idx])
        return self.scalar(v)

    # the same objective on ONE flat vector (the factors' entries one after the other, PackedEuclideanSpheres): no list of ~50 arrays is
    # taken apart and put together again around every evaluation
    def flat_layout(self, bounds):
        self._axes_pos = np.concatenate([np.arange(bounds[k], bounds[k + 1]) for k in self.axis_idx])
        self._scalar_pos = np.array([bounds[k] for k in self.scalar_idx])

    def _run_flat(self, v, want_grad):
        self.axes_host = np.ascontiguousarray(v[self._axes_pos])
        self.want_grad = want_grad
        return self.scalar(v[self._scalar_pos])

    def cost_flat(self, v):
        self.n_evals += 1
        loss, _ = self._run_flat(v, False)
        return float("inf") if loss >= 1e10 else float(loss)

    def egrad_flat(self, v):
        loss, g = self._run_flat(v, True)
        out = np.zeros(v.shape[0])
        out[self._scalar_pos] = g
        if loss < 1e10:
            out[self._axes_pos] = -self.grad_axes / self.y.numel()
        return out

    def cost(self, x):
        self.n_evals += 1
        loss, _ = self._run(x, False)
        return float("inf") if loss >= 1e10 else float(loss)

    def egrad(self, x):
        loss, g = self._run(x, True)
        n = self.y.numel()
        out = [None] * len(self.params)
        for pos, k in enumerate(self.scalar_idx):
            out[k] = np.full(np.shape(x[k]), g[pos])
        off = 0
        for level, k in enumerate(self.axis_idx):
            d = self.D - level
            out[k] = np.zeros(np.shape(x[k])) if loss >= 1e10 else (-self.grad_axes[off:off + d] / n).reshape(np.shape(x[k]))
            off += d
        return out


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
    problem = (_NestedSpdMllProblem.build(model, names, params, manifold) or _NestedSphereMllProblem.build(model, names, params, manifold)
               or _MllProblem(model, names, params, manifold))
    t1 = time.time()
    cands = [x0] if np.random.rand() < last_x_as_candidate_prob else []
    cands += [manifold.rand() for _ in range(nb_init_candidates - len(cands))]
    if keep_first_euclidean:
        eucl = [k for k, m in enumerate(factors) if isinstance(m, Euclidean)]
        for i in range(int(3 * nb_init_candidates / 4)):
            for k in eucl:
                cands[i][k] = x0[k].copy()
    problem.values_only = True           # (the candidates are only ranked: no gradient behind these values)
    costs = [problem.cost(c) for c in cands]
    problem.values_only = False
    x_init = cands[int(np.argmin(costs))]
    if isinstance(problem, _NestedSphereMllProblem) and all(type(m) in (Euclidean, Sphere) for m in factors):
        # one axis per nested level: the same product geometry on one flat vector (no Python loop over ~D factors per manifold operation)
        packed = PackedEuclideanSpheres(factors)

        problem.flat_layout(packed._bounds)

        class _Packed:
            manifold = packed
            cost = staticmethod(problem.cost_flat)
            grad = staticmethod(lambda v: packed.egrad2rgrad(v, problem.egrad_flat(v)))
        opt_v, log = solver.solve(_Packed, x=packed.pack(x_init))
        opt_x = [np.array(a) for a in packed.unpack(opt_v)]
    else:
        opt_x, log = solver.solve(problem, x=x_init)
    problem._set(opt_x)
    for p in params:
        p.grad = None
    info = {"fopt": problem.cost(opt_x), "wall_time": time.time() - t1, "opt_log": log, "init_cost": float(np.min(costs)),
            "cost_evals": problem.n_evals}
    model.invalidate()
    return model, info
