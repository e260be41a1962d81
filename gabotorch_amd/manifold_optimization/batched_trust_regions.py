"""Lock-step Riemannian trust regions over R restarts.

The reference solves the R restarts of the acquisition maximiser one after the other (manifold_optimize.py:207), each
with pymanopt-style trust regions + truncated CG (robust_trust_regions.py:111-570) or their constrained variant
(constrained_trust_regions.py:120-734).  Here the same state machine advances ALL restarts together: every cost /
gradient / Hessian-vector evaluation is one batched acquisition call (one kernel launch over R x n_train pairs), every
manifold operation one batched launch, and per-restart control flow is carried by boolean masks.  Restart r follows
exactly the arithmetic the reference's sequential solver would follow from the same initial point.

State lives in torch tensors on whatever device the problem's tensors live on; this file contains no device code.
"""
import time

import torch

NEGATIVE_CURVATURE, EXCEEDED_TR, REACHED_TARGET_LINEAR, REACHED_TARGET_SUPERLINEAR, MAX_INNER_ITER, MODEL_INCREASED, \
    REACHED_CONSTRAINTS = range(7)


def _library():
    from .. import _lib
    return _lib.load()


def _library_constraint_quick(con):
    """this package's own eigenvalue constraints bound by functools.partial: batch-safe by construction (and possibly under graph capture,
    where a host-side comparison could not run)"""
    try:
        from .manifold_optimize import _library_constraint
        return bool(_library_constraint(con))
    except Exception:       # noqa: BLE001
        return False


def _bm(mask, like):
    """broadcast a (R,) mask / scalar-per-restart tensor against (R, ...)"""
    return mask.reshape(mask.shape + (1,) * (like.dim() - mask.dim()))


def _randvec(man, x):
    """unit-norm random tangent vectors at the R points x ([3P] pymanopt `randvec`: a normal draw pushed into the tangent space and
    normalised in the manifold's metric); drawn from torch's global generator on x's device"""
    h = torch.randn_like(x)
    if hasattr(getattr(man, "base", man), "proj"):        # (_PointwiseManifold always has `proj`: ask the manifold it wraps)
        h = man.proj(x, h)
    else:
        h = man.egrad2rgrad(x, h)
    n = man.norm(x, h)
    return h / _bm(torch.where(n > 0, n, torch.ones_like(n)), h)


class BatchedProblem:
    """cost / Riemannian gradient / Riemannian Hessian-vector product for a batch of restarts.

    cost_fn(x: R x *shape) -> R tensor, differentiable by autograd (restarts are independent, so the gradient of the sum
    is the stack of per-restart gradients).  Mirrors pymanopt_addons/problem.py:14-159 + the PytorchBackend contract
    (tools/autodiff/_pytorch.py:83-116); `approx_hessian=True` is get_hessianfd (approximate_hessian.py:11-62)."""

    def __init__(self, manifold, cost_fn, approx_hessian=False, precon=None, use_hip_graphs=False, fused=None):
        self.manifold = manifold
        self.cost_fn = cost_fn
        # optional fused_acquisition.FusedAcquisition: value and first derivative as a fixed chain of HIP launches (no autograd);
        # cost_fn stays the definition and serves whatever the fused chain does not (exact Hessian-vector products)
        self.fused = fused
        self.approx_hessian = approx_hessian
        self.precon = precon or (lambda x, d: d)
        self.n_cost = 0
        self.n_grad = 0
        # A lock-step evaluation is a few hundred tiny launches (kernel + prep + GP algebra + their backward): launch bound.
        # With use_hip_graphs the value and value+gradient evaluations are captured once per input shape into hipGraphs
        # (torch.cuda.CUDAGraph) and replayed.  Needs a cost function without host synchronisation (error read-back off).
        self.use_hip_graphs = use_hip_graphs
        self._graphs = {}

    def _graphed(self, kind, x):
        key = (kind, tuple(x.shape), x.device)
        ent = self._graphs.get(key)
        if ent is None:
            static_x = x.detach().clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                for _ in range(3):                                  # warm-up outside capture (lazy caches, allocator)
                    self._eval(kind, static_x)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self._eval(kind, static_x)
            ent = (graph, static_x, outs)
            self._graphs[key] = ent
        graph, static_x, outs = ent
        static_x.copy_(x)
        graph.replay()
        return tuple(o.clone() for o in outs)

    def _eval(self, kind, x):
        if self.fused is not None:
            with torch.no_grad():
                return (self.fused.cost(x),) if kind == "cost" else self.fused.cost_egrad(x)
        if kind == "cost":
            with torch.no_grad():
                return (self.cost_fn(x).detach(),)
        xx = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            f = self.cost_fn(xx)
            (g,) = torch.autograd.grad(f.sum(), xx)
        return f.detach(), g.detach()

    def cost(self, x):
        self.n_cost += 1
        if self.use_hip_graphs and x.is_cuda and not getattr(self, "_capturing", False):
            return self._graphed("cost", x)[0]
        return self._eval("cost", x)[0]

    def cost_egrad(self, x, create_graph=False):
        self.n_grad += 1
        if self.use_hip_graphs and x.is_cuda and not create_graph and not getattr(self, "_capturing", False):
            f, g = self._graphed("grad", x)
            return f, g, None
        if self.fused is not None and not create_graph:
            f, g = self._eval("grad", x)
            return f, g, None
        xx = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            f = self.cost_fn(xx)
            (g,) = torch.autograd.grad(f.sum(), xx, create_graph=create_graph)
        return f.detach(), g, xx

    def cost_grad(self, x):
        f, eg, _ = self.cost_egrad(x)
        return f, self.manifold.egrad2rgrad(x, eg.detach())

    def grad(self, x):
        return self.cost_grad(x)[1]

    def hess(self, x, u, grad_x=None):
        man = self.manifold
        if self.approx_hessian:
            # finite difference of the gradient along u (approximate_hessian.py:30-60)
            norm_a = man.norm(x, u)
            g0 = self.grad(x) if grad_x is None else grad_x
            tiny = norm_a < 1e-15
            c = (2.0 ** -14) / torch.where(tiny, torch.ones_like(norm_a), norm_a)
            x1 = man.retr(x, _bm(c, u) * u)
            g1 = man.transp(x1, x, self.grad(x1))
            h = g1 / _bm(c, g1) - g0 / _bm(c, g0)
            return torch.where(_bm(tiny, h), torch.zeros_like(h), h)
        _, eg, xx = self.cost_egrad(x, create_graph=True)
        with torch.enable_grad():
            (eh,) = torch.autograd.grad((eg * u.detach()).sum(), xx)
        return man.ehess2rhess(x, eg.detach(), eh.detach(), u)


_batched_verdicts = {}       # id(callable) -> (callable, batch shape, dtype, takes a batch): BatchedTrustRegions._call_constraint


class _PointwiseManifold:
    """A manifold object whose methods take ONE point / tangent vector as numpy arrays (pymanopt's own classes) presented with
    batched torch signatures: every call walks the restarts on the host."""

    def __init__(self, manifold):
        self.base = manifold
        self._shape = getattr(manifold, "_shape", None)

    @property
    def dim(self):
        return self.base.dim

    @property
    def typicaldist(self):
        try:
            return self.base.typicaldist
        except NotImplementedError:
            return None

    def _map(self, name, like, *args):
        import numpy as np
        fn = getattr(self.base, name)
        rows = [fn(*[a[i].detach().cpu().numpy() for a in args]) for i in range(args[0].shape[0])]
        return torch.as_tensor(np.stack([np.asarray(r, dtype=np.float64) for r in rows])).to(like)

    def inner(self, x, u, v):
        return self._map("inner", x, x, u, v)

    def norm(self, x, u):
        return self._map("norm", x, x, u)

    def retr(self, x, u):
        return self._map("retr", x, x, u)

    def transp(self, x1, x2, d):
        return self._map("transp", x1, x1, x2, d)

    def egrad2rgrad(self, x, g):
        return self._map("egrad2rgrad", x, x, g)

    def ehess2rhess(self, x, eg, eh, u):
        return self._map("ehess2rhess", x, x, eg, eh, u)

    def proj(self, x, u):
        return self._map("proj", x, x, u)

    def rand(self):
        return self.base.rand()

    @staticmethod
    def zerovec(x):
        return torch.zeros_like(x)


class PointwiseProblemAdapter:
    """A pymanopt-style problem (`cost(x) -> float`, `grad(x)`, `hess(x, a)` on ONE point given as a numpy array - the object the
    reference hands to its solvers, pymanopt_addons/problem.py:14-159) presented with the BatchedProblem interface: every call
    walks the restarts on the host.  This is the route for user-supplied problems; the library's own acquisition problems are
    BatchedProblem instances and never come through here."""

    fused = None
    approx_hessian = False
    use_hip_graphs = False

    def __init__(self, problem):
        self.problem = problem
        # this package's manifolds (gabotorch_amd.manifolds) take batches of torch tensors natively; anything else is pymanopt's
        man = problem.manifold
        self.manifold = man if getattr(man, "batched", False) else _PointwiseManifold(man)
        self.n_cost = 0
        self.n_grad = 0

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy()

    def _stack(self, vals, like):
        import numpy as np
        return torch.as_tensor(np.stack([np.asarray(v, dtype=np.float64) for v in vals])).to(like)

    def cost(self, x):
        self.n_cost += 1
        return self._stack([self.problem.cost(self._np(xi)) for xi in x], x).reshape(x.shape[0])

    def cost_grad(self, x):
        self.n_grad += 1
        pts = [self._np(xi) for xi in x]
        f = self._stack([self.problem.cost(p) for p in pts], x).reshape(x.shape[0])
        return f, self._stack([self.problem.grad(p) for p in pts], x)

    def grad(self, x):
        return self.cost_grad(x)[1]

    def hess(self, x, u, grad_x=None):
        return self._stack([self.problem.hess(self._np(xi), self._np(ui)) for xi, ui in zip(x, u)], x)

    def precon(self, x, d):
        return self._stack([self.problem.precon(self._np(xi), self._np(di).copy()) for xi, di in zip(x, d)], x)


class BatchedTrustRegions:
    """Riemannian trust regions with truncated CG, all restarts in lock step; optional equality / inequality constraints
    handled as in ConstrainedTrustRegions (linearised constraints truncate the tCG step at distance Delta_cons).

    Stopping criteria: `mingradnorm` and `maxiter` per restart on every path.  `maxtime` is checked once per outer iteration by the host-
    driven paths; the single-launch device solve cannot look at a clock, so it is only taken when `maxtime` is at its default (>= 1000 s)
    and a caller that lowers `maxtime` gets the launch-per-iteration plan instead.  pymanopt's underscored attribute names (`_maxiter`,
    `_mingradnorm`, ...) are aliases of the plain ones, so outer solvers that tighten them between calls act on this object."""

    def __init__(self, miniter=3, kappa=0.1, theta=1.0, rho_prime=0.1, use_rand=False, rho_regularization=1e3, maxtime=1000,
                 maxiter=1000, mingradnorm=1e-6, minstepsize=1e-10, maxcostevals=5000, logverbosity=0, strict_constraints=False):
        # strict_constraints=True = StrictConstrainedTrustRegions (constrained_trust_regions.py:737-1415): a proposal that
        # violates a constraint is rejected outright (cost = +inf) and the radius shrinks; everything else is identical.
        self.strict_constraints = strict_constraints
        # use_rand=True (robust_trust_regions.py:173-219, 407-452): tCG starts from a tiny random tangent vector instead of zero, runs
        # without the preconditioner, and its result is compared with the Cauchy point.  On S^d_++ with the fused acquisition the
        # device-resident tCG launches take it (gabo_spd_tcg_begin_rand); everywhere else the generic lock-step path.
        self.use_rand = bool(use_rand)
        self.miniter, self.kappa, self.theta, self.rho_prime = miniter, kappa, theta, rho_prime
        self.rho_regularization = rho_regularization
        self.maxtime, self.maxiter, self.mingradnorm = maxtime, maxiter, mingradnorm
        self.minstepsize, self.maxcostevals = minstepsize, maxcostevals
        self.logverbosity = logverbosity
        self.log = {}
        # set to a list to record, per outer iteration of the generic lock-step path, the state the reference's solvers go through
        # (iterate, radius, tCG step / stop reason, ratio, acceptance): what tests/golden/tr_traces.npz holds for the reference
        self.trace = None

    # pymanopt's Solver keeps its stopping criteria under underscored names and outer solvers (e.g. an augmented Lagrangian loop)
    # tighten `_mingradnorm` between calls: same storage here
    _maxtime = property(lambda self: self.maxtime, lambda self, v: setattr(self, "maxtime", v))
    _maxiter = property(lambda self: self.maxiter, lambda self, v: setattr(self, "maxiter", v))
    _mingradnorm = property(lambda self: self.mingradnorm, lambda self, v: setattr(self, "mingradnorm", v))
    _minstepsize = property(lambda self: self.minstepsize, lambda self, v: setattr(self, "minstepsize", v))
    _maxcostevals = property(lambda self: self.maxcostevals, lambda self, v: setattr(self, "maxcostevals", v))
    _logverbosity = property(lambda self: self.logverbosity, lambda self, v: setattr(self, "logverbosity", v))

    # ------------------------------------------------------------------------------------------------- constraints
    @staticmethod
    def _nested_group(x, constraints):
        """(is_max, bounds, (w, x0, p)) when EVERY constraint is a functools.partial over max/min_eigenvalue_nested_spd_constraint with
        one common mapping (HD-GaBO's bounds in the original space): one launch of gabo_nested_spd_extreme_eigenvalues then serves them
        all - both extreme eigenpairs come out of the same Householder reduction - with their gradients, no autograd graph.  None
        otherwise (the callables are called one by one)."""
        if not constraints or not torch.is_tensor(x) or not x.is_cuda:
            return None
        from .. import _lib
        from ..Riemannian_utils.spd_constraints_utils_torch import builtin_constraint
        info = [builtin_constraint(c) for c in constraints]
        if any(b is None or len(b) != 3 for b in info):
            return None
        first = info[0][2]
        if any(any(s is not t for s, t in zip(first, b[2])) for b in info[1:]):
            return None
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in first)
        memo = getattr(first[2], "_gabo_lift", None)         # (the memo nested_spd_constraints_utils._lifted_extremes keeps)
        if memo is None or memo[0] != key:
            from .. import ops
            memo = (key, ops.nested_spd_lift_prepare(*first))
            try:
                first[2]._gabo_lift = memo
            except AttributeError:
                pass
        return [b[0] == _lib.GABO_CONSTRAINT_MAX_EIGENVALUE_NESTED for b in info], [b[1] for b in info], memo[1]

    @staticmethod
    def _call_constraint(con, x):
        """Values (R,) of one user constraint at the R points x.  The reference's convention is a callable of ONE point (`x[1] - yc`,
        gabo_sphere_equality_constraints.py:106-107); a callable written for a batch (`x[..., 1] - yc`) saves R - 1 calls.  The batched call is
        accepted only if it has the batch's shape AND agrees with the single-point call at the first and the last restart: a one-point
        callable applied to the batch can return the right SHAPE with the wrong meaning (`x[1] - yc` on an R x dim batch is row 1, of length
        dim - which passes a shape test whenever the number of restarts equals the dimension)."""
        R = x.shape[0]
        # the verdict on a callable is the same every time it is asked for this batch shape and dtype: validated once (two single-point calls and
        # a host comparison), remembered with the callable itself (no id() reuse); the solvers ask once per constraint per outer iteration
        held = _batched_verdicts.get(id(con))
        known = held[3] if (held is not None and held[0] is con and held[1] == tuple(x.shape) and held[2] == x.dtype) else None
        if known is False:
            return torch.stack([torch.as_tensor(con(x[i])).reshape(()) for i in range(R)])
        try:
            f = con(x)
            ok = torch.is_tensor(f) and f.shape == x.shape[:1]
        except (IndexError, ValueError, TypeError, RuntimeError):
            ok = False
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()      # (capture_constraints=True: the caller vouches for the callable)
        if ok and known is None and not capturing and not _library_constraint_quick(con):
            with torch.no_grad():
                ends = torch.stack([torch.as_tensor(con(x[i])).reshape(()) for i in sorted({0, R - 1})]).to(f.dtype)
                got = f.detach()[sorted({0, R - 1})]
                # (a batched callable may reduce in another order than the one-point call: the tolerance follows the dtype)
                eps = torch.finfo(f.dtype).eps if f.dtype.is_floating_point else 0.0
                ok = bool(torch.allclose(got, ends, rtol=max(1e-9, 64 * eps), atol=max(1e-12, 64 * eps), equal_nan=True))
        if not capturing and known is None:
            if len(_batched_verdicts) > 256:
                _batched_verdicts.clear()
            _batched_verdicts[id(con)] = (con, tuple(x.shape), x.dtype, bool(ok))
        if not ok:
            # one point at a time (a genuine error in the callable is raised again by these calls)
            f = torch.stack([torch.as_tensor(con(x[i])).reshape(()) for i in range(R)])
        return f

    @staticmethod
    def _constraint_values_grads(problem, x, constraints):
        """-> fc (R, C), rgrad (C tensors of shape R x *shape)"""
        group = BatchedTrustRegions._nested_group(x, constraints)
        if group is not None:
            from .. import ops
            is_max, bounds, (w, x0, p) = group
            lam, grad = ops.nested_spd_extreme_eigenvalues(x.detach(), w, p, x0, want_grad=True)
            vals = [(b - lam[:, 0]) if mx else (lam[:, 1] - b) for mx, b in zip(is_max, bounds)]
            grads = [problem.manifold.egrad2rgrad(x, (-grad[:, 0]) if mx else grad[:, 1]) for mx in is_max]
            return torch.stack(vals, dim=1).to(x.dtype), grads
        vals, grads = [], []
        for con in constraints:
            xx = x.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                f = BatchedTrustRegions._call_constraint(con, xx)
                (g,) = torch.autograd.grad(f.sum(), xx, allow_unused=True)
            if g is None:
                g = torch.zeros_like(x)
            vals.append(f.detach().to(x.dtype))
            grads.append(problem.manifold.egrad2rgrad(x, g.detach()))
        return torch.stack(vals, dim=1), grads

    @staticmethod
    def _constraint_values(x, constraints):
        """-> fc (R, C) only (the strict variant's feasibility test of a proposal needs no gradients)"""
        group = BatchedTrustRegions._nested_group(x, constraints)
        if group is not None:
            from .. import ops
            is_max, bounds, (w, x0, p) = group
            lam = ops.nested_spd_extreme_eigenvalues(x.detach(), w, p, x0)
            return torch.stack([(b - lam[:, 0]) if mx else (lam[:, 1] - b) for mx, b in zip(is_max, bounds)], dim=1).to(x.dtype)
        vals = []
        with torch.no_grad():
            for con in constraints:
                vals.append(BatchedTrustRegions._call_constraint(con, x).detach().to(x.dtype))
        return torch.stack(vals, dim=1)

    # ------------------------------------------------------------------------------------------------- solve
    def solve(self, problem, x=None, eq_constraints=None, ineq_constraints=None, mininner=1, maxinner=None, Delta_bar=None,
              Delta0=None, Delta_cons=None):
        """The reference solvers' entry point (robust_trust_regions.py:111-112, constrained_trust_regions.py:120-121, :783-784).

        Lock-step form: `problem` a BatchedProblem, `x` a tensor R x *point_shape of initial points -> the R optimised points.
        pymanopt form: `problem` exposes cost/grad/hess on single numpy points (pymanopt_addons Problem) and `x` is ONE point (numpy
        array or tensor; None draws `manifold.rand()`) -> the optimised point as the same kind of object.  Both run the same state
        machine; the second evaluates the problem's callables restart by restart on the host."""
        import numpy as np
        batched = isinstance(problem, BatchedProblem) or hasattr(problem, "cost_grad")
        if not batched:
            problem = PointwiseProblemAdapter(problem)
        if x is None:
            x = problem.manifold.rand()
        # ONE point: always for a pymanopt-style problem or numpy input; for a batched problem only when the tensor has exactly the
        # manifold's point shape (an R x d_vec batch of Mandel rows on S^n_++ has the RANK of one n x n point, not its shape)
        shape = getattr(problem.manifold, "_shape", None)
        single = not batched or not torch.is_tensor(x) or (shape is not None and tuple(x.shape) == tuple(shape)) \
            or (shape is None and x.dim() == 1)
        if single:
            was_numpy = not torch.is_tensor(x)
            xt = torch.as_tensor(np.asarray(x, dtype=np.float64)) if was_numpy else x.detach().double()
            out = self._solve(problem, xt[None], eq_constraints, ineq_constraints, mininner, maxinner, Delta_bar, Delta0, Delta_cons)[0]
            out = out.cpu().numpy() if was_numpy else out
            # pymanopt's contract (robust_trust_regions.py:393-398): (x, optlog) when logverbosity >= 1
            return (out, self.log) if (self.logverbosity >= 1 and not batched) else out
        return self._solve(problem, x, eq_constraints, ineq_constraints, mininner, maxinner, Delta_bar, Delta0, Delta_cons)

    def _solve(self, problem, x, eq_constraints, ineq_constraints, mininner, maxinner, Delta_bar, Delta0, Delta_cons):
        """x: R x *point_shape initial points.  Returns the R optimised points."""
        man = problem.manifold
        x = x.detach().clone()
        R = x.shape[0]
        dt, dev = x.dtype, x.device
        if maxinner is None:
            maxinner = man.dim
        if Delta_bar is None:
            Delta_bar = getattr(man, "typicaldist", None) or float(man.dim) ** 0.5
        if Delta0 is None:
            Delta0 = Delta_bar / 8
        if Delta_cons is None:
            Delta_cons = 1e-6
        eqs = list(eq_constraints) if isinstance(eq_constraints, (list, tuple)) else ([eq_constraints] if eq_constraints else [])
        ineqs = list(ineq_constraints) if isinstance(ineq_constraints, (list, tuple)) else ([ineq_constraints] if ineq_constraints else [])
        neq = len(eqs)
        constrained = bool(eqs or ineqs)

        # use_rand: the SPD plan of device-resident tCG launches takes it (gabo_spd_tcg_begin_rand); the single-launch kernels and
        # the sphere's do not - no reference caller sets it - and fall through to the generic lock-step code below
        rand_ok = not self.use_rand or getattr(getattr(problem, "fused", None), "family", None) == "spd"
        if rand_ok and self._device_tcg_applies(problem, x, len(eqs) + len(ineqs)) and getattr(problem, "device_outer", True):
            return self._solve_device(problem, x, eqs, ineqs, mininner, maxinner, Delta_bar, Delta0, Delta_cons)
        time0 = time.time()
        fx, g = problem.cost_grad(x)
        ng = man.norm(x, g)
        Delta = torch.full((R,), float(Delta0), dtype=dt, device=dev)
        active = torch.ones(R, dtype=torch.bool, device=dev)
        iters = torch.zeros(R, dtype=torch.long, device=dev)
        k = 0
        eps = torch.finfo(dt).eps
        while True:
            if constrained:
                fc, gc = self._constraint_values_grads(problem, x, eqs + ineqs)
            else:
                fc, gc = None, []
            eta0 = None
            if self.use_rand:                                             # (robust_trust_regions.py:176-181)
                eta0 = 1e-6 * _randvec(man, x)
                for _ in range(64):
                    big = man.norm(x, eta0) > Delta
                    if not bool(big.any()):
                        break
                    eta0 = torch.where(_bm(big, eta0), eta0 * float(eps) ** 0.25, eta0)
            eta, Heta, stop_inner = self._tcg(problem, x, g, Delta, active, mininner, maxinner, fc, gc, neq, Delta_cons, eta0=eta0)
            if self.use_rand:
                # keep the better of the tCG step and the Cauchy point (:196-219)
                Hg = problem.hess(x, g, grad_x=g)
                g_Hg = man.inner(x, g, Hg)
                safe = torch.where(g_Hg > 0, g_Hg, torch.ones_like(g_Hg))
                tau_c = torch.where(g_Hg <= 0, torch.ones_like(ng), torch.clamp(ng ** 3 / (Delta * safe), max=1.0))
                scale = -tau_c * Delta / torch.where(ng > 0, ng, torch.ones_like(ng))
                eta_c, Heta_c = _bm(scale, g) * g, _bm(scale, Hg) * Hg
                mdle = fx + man.inner(x, g, eta) + 0.5 * man.inner(x, Heta, eta)
                mdlec = fx + man.inner(x, g, eta_c) + 0.5 * man.inner(x, Heta_c, eta_c)
                cauchy = mdlec < mdle
                eta = torch.where(_bm(cauchy, eta), eta_c, eta)
                Heta = torch.where(_bm(cauchy, Heta), Heta_c, Heta)
            x_prop = man.retr(x, eta)
            fx_prop = problem.cost(x_prop)
            invalid = torch.zeros_like(active)
            if constrained and self.strict_constraints:
                fcp = self._constraint_values(x_prop, eqs + ineqs)                          # (:932-951)
                viol = fcp.clone()
                viol[:, neq:] = torch.clamp(viol[:, neq:], max=0.0)
                invalid = viol.abs().sum(1) != 0
                fx_prop = torch.where(invalid, torch.full_like(fx_prop, float("inf")), fx_prop)
            rhonum = fx - fx_prop
            rhoden = -man.inner(x, g, eta) - 0.5 * man.inner(x, eta, Heta)
            rho_reg = torch.clamp(fx.abs(), min=1.0) * eps * self.rho_regularization
            rhonum = rhonum + rho_reg
            rhoden = rhoden + rho_reg
            model_decreased = rhoden >= 0
            rho = torch.where(rhoden == 0, torch.full_like(rhoden, float("nan")), rhonum / rhoden)
            shrink = (rho < 0.25) | ~model_decreased | torch.isnan(rho) | invalid
            boundary = (stop_inner == NEGATIVE_CURVATURE) | (stop_inner == EXCEEDED_TR)
            if constrained:
                boundary = boundary | (stop_inner == REACHED_CONSTRAINTS)
            grow = ~shrink & (rho > 0.75) & boundary
            newDelta = torch.where(shrink, Delta / 4, torch.where(grow, torch.clamp(2 * Delta, max=float(Delta_bar)), Delta))
            Delta_before = Delta
            Delta = torch.where(active, newDelta, Delta)
            accept = active & model_decreased & (rho > self.rho_prime)
            if self.trace is not None:
                self.trace.append({"x": x.clone(), "Delta": Delta_before, "eta": eta.clone(), "stop_inner": stop_inner.clone(),
                                   "rho": rho.clone(), "accept": accept.clone(), "active": active.clone(), "fx": fx.clone()})
            if bool(accept.any()):
                x = torch.where(_bm(accept, x), x_prop, x)
                fx = torch.where(accept, fx_prop, fx)
                _, gnew = problem.cost_grad(x)
                g = torch.where(_bm(accept, g), gnew, g)
                ng = torch.where(accept, man.norm(x, g), ng)
            k += 1
            iters = iters + active.long()
            stop = (ng < self.mingradnorm) | (iters >= self.maxiter)
            active = active & ~stop
            if not bool(active.any()) or (time.time() - time0) >= self.maxtime:
                break
        self.log = {"iterations": k, "per_restart_iterations": iters, "final_cost": fx, "final_gradnorm": ng,
                    "cost_evals": problem.n_cost, "grad_evals": problem.n_grad, "time": time.time() - time0}
        return x

    # ------------------------------------------------------------------------------------------------- device-resident solve
    def _solve_device(self, problem, x, eqs, ineqs, mininner, maxinner, Delta_bar, Delta0, Delta_cons):
        """solve() for the case _device_tcg_applies: one trust-region iteration is a FIXED sequence of launches on persistent
        buffers with per-restart masks (no data-dependent host control flow), so with hipGraphs it is captured once and each outer
        iteration is one replay + one read-back of "any restart still active".  Same arithmetic per restart as solve()."""
        from .. import ops
        man, fused = problem.manifold, problem.fused
        R, d = x.shape[0], x.shape[-1]
        dt, dev = x.dtype, x.device
        cons = eqs + ineqs
        ncons, neq = len(cons), len(eqs)
        sphere = fused.family == "sphere"
        # (the random start is drawn between the launches; a trace clones the state between them)
        graphs = bool(getattr(problem, "use_hip_graphs", False)) and not self.use_rand and self.trace is None
        # Will the whole solve be ONE launch (decided below by the same tests)?  Then the tCG handle, the evaluation buffers and the
        # constraint buffers of the multi-launch plans are never touched: not creating them takes ~10 allocations and fill launches
        # (~0.1 ms of host time) off the front of a 4-ms sweep.
        one_launch = False
        builtins, lift = None, None
        fused_kernels = getattr(fused, "single_launch", False) and getattr(problem, "device_iteration", True) and not self.use_rand
        # (the library says which of its iteration kernels exist for this surrogate: e.g. no propose / update pair for the log-Euclidean
        # surrogate at d = 8, csrc/spd_tr_le_hi.hip, and no generic-workspace single launch for it at d = 7, 8, csrc/spd_tr_body.hpp)
        propose_ok = bool(fused_kernels and (sphere or _library().gabo_spd_tr_propose_supported(int(fused.mode) | int(fused.metric), d)))
        if fused_kernels:
            from ..Riemannian_utils.spd_constraints_utils_torch import builtin_constraint, builtin_lift
            builtins = [builtin_constraint(c) for c in cons]
            solve_ok = (ncons == 0) if sphere else (d <= 8 and neq == 0 and all(b is not None for b in builtins))
            lift = builtin_lift(builtins) if (solve_ok and not sphere) else None      # the nested kinds' mapping (one for all of them)
            solve_ok = solve_ok and lift is not False
            if solve_ok and not sphere:
                # (the library's own word: e.g. no single-launch form of the log-Euclidean surrogate at d = 7, 8 beyond what its LDS holds)
                import ctypes
                solve_ok = bool(_library().gabo_spd_tr_solve_supported(ctypes.byref(fused.acq_params()), R, d, ncons,
                                                                       0 if lift is None else int(lift[0].shape[0])))
            one_launch = bool(solve_ok and getattr(problem, "device_solve", True) and self.maxtime >= 1000)
        fused_iteration = propose_ok or one_launch
        T = None if (sphere or one_launch) else ops.SpdTcg(R, d, ncons, dev)
        val_buf = None if one_launch else torch.zeros(R, dtype=dt, device=dev)
        eg_buf = None if one_launch else torch.zeros(R, d * (d + 1) // 2, dtype=dt, device=dev)
        eps = torch.finfo(dt).eps
        time0 = time.time()
        fx, eg = fused.cost_egrad(x)
        problem.n_grad += 1

        class S:      # persistent state, updated in place
            pass
        S.x, S.fx = x.detach().clone(), fx.clone()
        S.g = man.egrad2rgrad(S.x, eg)
        S.ng = man.norm(S.x, S.g)
        S.Delta = torch.full((R,), float(Delta0), dtype=dt, device=dev)
        S.active = torch.ones(R, dtype=torch.bool, device=dev)
        S.iters = torch.zeros(R, dtype=torch.long, device=dev)
        S.any_active = None if one_launch else torch.ones((), dtype=torch.bool, device=dev)
        step_args = (neq, Delta_cons, self.theta, self.kappa, mininner)

        fc_buf = torch.zeros(R, ncons, dtype=dt, device=dev) if (ncons and not one_launch) else None
        gc_buf = torch.zeros((ncons,) + tuple(x.shape), dtype=dt, device=dev) if (ncons and not one_launch) else None
        invalid_buf = None if one_launch else torch.zeros(R, dtype=torch.bool, device=dev)
        strict = bool(ncons and self.strict_constraints)

        def constraints_at_x():                      # user callables (torch): captured only on request, see below
            fc, gc = self._constraint_values_grads(problem, S.x, cons)
            fc_buf.copy_(fc)
            gc_buf.copy_(torch.stack(gc))

        def constraints_at_proposal(A):              # StrictConstrainedTrustRegions (constrained_trust_regions.py:932-951)
            fcp = self._constraint_values(A["x_prop"], cons)
            viol = fcp.clone()
            viol[:, neq:] = torch.clamp(viol[:, neq:], max=0.0)
            invalid_buf.copy_(viol.abs().sum(1) != 0)

        def part_a(sync):
            T.begin(S.x, S.g, gc_buf, fc_buf, S.active, S.Delta)
            if self.use_rand:                                             # (robust_trust_regions.py:176-181, 411-415: as _solve / _tcg_begin)
                eta0 = 1e-6 * _randvec(man, S.x)
                for _ in range(64):
                    big = man.norm(S.x, eta0) > S.Delta
                    if not bool(big.any()):
                        break
                    eta0 = torch.where(_bm(big, eta0), eta0 * float(eps) ** 0.25, eta0)
                T.begin_rand(eta0, problem.hess(S.x, eta0, grad_x=S.g))
            for _ in range(int(maxinner)):
                T.step(fused.egrad_mandel(T.fd_point(), active_ptr=T.running_ptr, out=(val_buf, eg_buf)), *step_args)
                problem.n_grad += 1
                if sync and not bool(T.any_running.item()):
                    break
            eta, Heta, stop_inner = T.end()
            if self.use_rand:
                # keep the better of the tCG step and the Cauchy point (:196-219), the statements of _solve
                Hg = problem.hess(S.x, S.g, grad_x=S.g)
                g_Hg = man.inner(S.x, S.g, Hg)
                safe = torch.where(g_Hg > 0, g_Hg, torch.ones_like(g_Hg))
                tau_c = torch.where(g_Hg <= 0, torch.ones_like(S.ng), torch.clamp(S.ng ** 3 / (S.Delta * safe), max=1.0))
                scale = -tau_c * S.Delta / torch.where(S.ng > 0, S.ng, torch.ones_like(S.ng))
                eta_c, Heta_c = _bm(scale, S.g) * S.g, _bm(scale, Hg) * Hg
                mdle = S.fx + man.inner(S.x, S.g, eta) + 0.5 * man.inner(S.x, Heta, eta)
                mdlec = S.fx + man.inner(S.x, S.g, eta_c) + 0.5 * man.inner(S.x, Heta_c, eta_c)
                cauchy = mdlec < mdle
                eta = torch.where(_bm(cauchy, eta), eta_c, eta)
                Heta = torch.where(_bm(cauchy, Heta), Heta_c, Heta)
            x_prop = man.retr(S.x, eta)
            fx_prop, eg_prop = fused.cost_egrad(x_prop)
            problem.n_grad += 1
            return {"eta": eta, "Heta": Heta, "stop_inner": stop_inner, "x_prop": x_prop, "fx_prop": fx_prop, "eg_prop": eg_prop}

        def part_b(A):
            eta, Heta, stop_inner, x_prop, eg_prop = A["eta"], A["Heta"], A["stop_inner"], A["x_prop"], A["eg_prop"]
            fx_prop = A["fx_prop"]
            invalid = invalid_buf
            if strict:
                fx_prop = torch.where(invalid, torch.full_like(fx_prop, float("inf")), fx_prop)
            rhonum = S.fx - fx_prop
            rhoden = -man.inner(S.x, S.g, eta) - 0.5 * man.inner(S.x, eta, Heta)
            rho_reg = torch.clamp(S.fx.abs(), min=1.0) * eps * self.rho_regularization
            rhonum = rhonum + rho_reg
            rhoden = rhoden + rho_reg
            model_decreased = rhoden >= 0
            rho = torch.where(rhoden == 0, torch.full_like(rhoden, float("nan")), rhonum / rhoden)
            shrink = (rho < 0.25) | ~model_decreased | torch.isnan(rho) | invalid
            boundary = (stop_inner == NEGATIVE_CURVATURE) | (stop_inner == EXCEEDED_TR)
            if ncons:
                boundary = boundary | (stop_inner == REACHED_CONSTRAINTS)
            grow = ~shrink & (rho > 0.75) & boundary
            newDelta = torch.where(shrink, S.Delta / 4, torch.where(grow, torch.clamp(2 * S.Delta, max=float(Delta_bar)), S.Delta))
            accept = S.active & model_decreased & (rho > self.rho_prime)
            if self.trace is not None:                                  # (the record of _solve, same keys)
                self.trace.append({"x": S.x.clone(), "Delta": S.Delta.clone(), "eta": eta.clone(), "stop_inner": stop_inner.clone(),
                                   "rho": rho.clone(), "accept": accept.clone(), "active": S.active.clone(), "fx": S.fx.clone()})
            S.Delta.copy_(torch.where(S.active, newDelta, S.Delta))
            gnew = man.egrad2rgrad(x_prop, eg_prop)                 # the gradient at the proposal IS the gradient at the new x
            S.x.copy_(torch.where(_bm(accept, S.x), x_prop, S.x))
            S.fx.copy_(torch.where(accept, fx_prop, S.fx))
            S.g.copy_(torch.where(_bm(accept, S.g), gnew, S.g))
            S.ng.copy_(torch.where(accept, man.norm(S.x, S.g), S.ng))
            S.iters.add_(S.active.long())
            stop = (S.ng < self.mingradnorm) | (S.iters >= self.maxiter)
            S.active.copy_(S.active & ~stop)
            S.any_active.copy_(S.active.any())

        # d <= 12: the two parts are ONE launch each (csrc/spd_tr.hip: every wave runs its restart's whole tCG loop, proposal and
        # acquisition evaluations by itself)
        if fused_iteration:
            if sphere:
                TR = ops.SphereTr(R, d, ncons, fused.sphere_acq_params(), dev, exact_hessian=not problem.approx_hessian)
            else:
                TR = ops.SpdTr(R, d, ncons, fused.acq_params(), fused.train.shape[0], dev)
            S.active_u8 = S.active.view(torch.uint8)
            inv_u8 = None if invalid_buf is None else invalid_buf.view(torch.uint8)

            def part_a(sync):       # noqa: F811
                xp = TR.propose(S.x, S.g, S.Delta, S.active_u8, gc_buf, fc_buf, neq, Delta_cons, self.theta, self.kappa, mininner,
                                maxinner)
                return {"x_prop": xp}

            # (where the record below finds the stop reasons of the last tCG run; TcgWs: stop, then running)
            stop_off = (_library().gabo_sphere_tr_stop_offset(R, d, ncons) // 4 if sphere
                        else _library().gabo_spd_tcg_running_offset(R, d, ncons) // 4 - R)

            def part_b(A):          # noqa: F811
                if self.trace is not None:
                    self.trace.append({"x": S.x.clone(), "Delta": S.Delta.clone(), "active": S.active.clone(), "fx": S.fx.clone(),
                                       "stop_inner": TR.ws.view(torch.int32)[stop_off:stop_off + R].clone().long()})
                TR.update(S.x, S.fx, S.g, S.ng, S.Delta, S.active_u8, S.iters, inv_u8 if strict else None, Delta_bar, self.rho_prime,
                          self.rho_regularization, self.mingradnorm, self.maxiter)
                S.any_active.copy_(TR.any_active[0] != 0)

            # no constraint needs a host callable (none, or eigenvalue bounds built with functools.partial as in the reference
            # examples): the whole solve is ONE launch, every wave iterating its restart to the end
            if one_launch:
                extra = {} if sphere else {"lift": lift}
                if self.trace is not None:
                    # the launch writes its own record (gabo_tr_solve_record): iterate, radius, tCG stop reason per outer iteration
                    L = d if sphere else d * d
                    extra["record"] = torch.full((int(min(self.maxiter, 1 << 14)), R, L + 2), float("nan"), dtype=dt, device=dev)
                TR.solve(S.x, S.fx, S.g, S.ng, S.Delta, S.active_u8, S.iters, [b[0] for b in builtins], [b[1] for b in builtins], strict,
                          Delta_cons, self.theta, self.kappa, mininner, maxinner, Delta_bar, self.rho_prime, self.rho_regularization,
                          self.mingradnorm, self.maxiter, **extra)
                if hasattr(TR, "status"):
                    ops._raise_if_not_spd(TR.status, "gabo_spd_tr_solve")       # (when error checking is on: one read-back per solve)
                k = int(S.iters.max().item())
                if self.trace is not None:
                    rec = extra["record"]
                    for kk in range(min(k, rec.shape[0])):
                        ran = ~torch.isnan(rec[kk, :, L])
                        self.trace.append({"x": rec[kk, :, :L].reshape(x.shape).clone(), "Delta": rec[kk, :, L].clone(), "active": ran,
                                           "stop_inner": torch.where(ran, rec[kk, :, L + 1], torch.full_like(rec[kk, :, L], -1.0)).long()})
                ops.check_deferred()
                self.log = {"iterations": k, "per_restart_iterations": S.iters, "final_cost": S.fx, "final_gradnorm": S.ng,
                            "cost_evals": problem.n_cost, "grad_evals": problem.n_grad, "time": time.time() - time0, "one_launch_solve": True}
                return S.x

        # Execution plan.  Eager: the parts in order, with the inner loop leaving as soon as no restart runs.  hipGraphs: the
        # launches between two evaluations of the USER's constraint callables form one graph; the callables themselves run
        # eagerly between replays (they may synchronise) unless the caller vouches for them with capture_constraints=True.
        capture_cons = bool(getattr(problem, "capture_constraints", False))
        plan = []               # list of callables executed once per outer iteration
        prev_check = None
        if not graphs:
            holder = {}
            if ncons:
                plan.append(constraints_at_x)
            plan.append(lambda: holder.update(part_a(True)))
            if strict:
                plan.append(lambda: constraints_at_proposal(holder))
            plan.append(lambda: part_b(holder))
        else:
            prev_check = ops.set_error_checking(False)       # a status read-back is a host sync: not capturable
            pool = None

            def capture(fn):
                nonlocal pool
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, pool=pool):
                    out = fn()
                pool = gr.pool()
                return gr, out

            if not ncons or capture_cons:
                def whole():
                    if ncons:
                        constraints_at_x()
                    A = part_a(False)
                    if strict:
                        constraints_at_proposal(A)
                    part_b(A)
                gr, _ = capture(whole)
                plan.append(gr.replay)
            elif not strict:
                gr, _ = capture(lambda: part_b(part_a(False)))
                plan += [constraints_at_x, gr.replay]
            else:
                ga, A = capture(lambda: part_a(False))
                gb, _ = capture(lambda: part_b(A))
                plan += [constraints_at_x, ga.replay, lambda: constraints_at_proposal(A), gb.replay]
        k = 0
        try:
            while True:
                for stage in plan:
                    stage()
                k += 1
                if not bool(S.any_active) or (time.time() - time0) >= self.maxtime:
                    break
        finally:
            if prev_check is not None:
                ops.set_error_checking(prev_check)
        self.log = {"iterations": k, "per_restart_iterations": S.iters, "final_cost": S.fx, "final_gradnorm": S.ng,
                    "cost_evals": problem.n_cost, "grad_evals": problem.n_grad, "time": time.time() - time0}
        return S.x

    # ------------------------------------------------------------------------------------------------- truncated CG
    @staticmethod
    def _device_tcg_applies(problem, x, ncons):
        """SPD manifold + fused acquisition chain + FD Hessian + the reference preconditioner: the whole tCG runs in HIP kernels."""
        from ..manifolds import PositiveDefinite, Sphere
        fused = getattr(problem, "fused", None)
        if (fused is None or not x.is_cuda or not getattr(problem, "reference_precon", False)
                or not getattr(problem, "device_tcg", True) or ncons > 8 or x.dtype != torch.float64):
            return False
        if fused.family == "sphere":       # csrc/sphere_tr.hip: propose / update / solve kernels, FD or exact (closed-form) Hessian
            return (isinstance(problem.manifold, Sphere) and fused.single_launch and x.dim() == 2
                    and getattr(problem, "device_iteration", True))
        return (problem.approx_hessian and fused.family == "spd" and fused.matrix_input
                and isinstance(problem.manifold, PositiveDefinite) and x.shape[-1] <= 32)

    class _TcgState:
        """All tCG quantities of the R restarts as persistent tensors updated IN PLACE, so that one iteration is a fixed sequence
        of launches on fixed buffers - which is what lets it be captured in a hipGraph and replayed."""

        def __init__(self, x, ncons):
            R, dt, dev = x.shape[0], x.dtype, x.device
            z1 = lambda: torch.zeros(R, dtype=dt, device=dev)       # noqa: E731
            self.x, self.g = torch.zeros_like(x), torch.zeros_like(x)
            self.Delta, self.active = z1(), torch.zeros(R, dtype=torch.bool, device=dev)
            self.eta, self.Heta, self.r, self.delta = (torch.zeros_like(x) for _ in range(4))
            self.e_Pe, self.e_Pd, self.d_Pd, self.z_r, self.model_value, self.norm_r0 = (z1() for _ in range(6))
            self.stop = torch.zeros(R, dtype=torch.long, device=dev)
            self.running = torch.zeros(R, dtype=torch.bool, device=dev)
            self.any_running = torch.zeros((), dtype=torch.bool, device=dev)
            self.fc = torch.zeros(R, ncons, dtype=dt, device=dev) if ncons else None
            self.fcg_Pe = torch.zeros(R, ncons, dtype=dt, device=dev) if ncons else None
            self.gc = [torch.zeros_like(x) for _ in range(ncons)]

    def _tcg_begin(self, problem, S, eta0=None):
        man = problem.manifold
        if eta0 is None:
            S.eta.zero_()
            S.Heta.zero_()
            S.r.copy_(S.g)
            S.e_Pe.zero_()
        else:                                       # use_rand: eta0 ~ 0 given by the caller, no preconditioner (:411-415)
            S.eta.copy_(eta0)
            S.Heta.copy_(problem.hess(S.x, S.eta, grad_x=S.g))
            S.r.copy_(S.g + S.Heta)
            S.e_Pe.copy_(man.inner(S.x, S.eta, S.eta))
        S.norm_r0.copy_(man.inner(S.x, S.r, S.r).clamp(min=0).sqrt())
        z = problem.precon(S.x, S.r) if eta0 is None else S.r
        S.z_r.copy_(man.inner(S.x, z, S.r))
        S.d_Pd.copy_(S.z_r)
        S.delta.copy_(-z)
        if eta0 is None:
            S.e_Pd.zero_()
            S.model_value.zero_()
        else:
            S.e_Pd.copy_(man.inner(S.x, S.eta, S.delta))
            S.model_value.copy_(man.inner(S.x, S.eta, S.g) + 0.5 * man.inner(S.x, S.eta, S.Heta))
        S.stop.fill_(MAX_INNER_ITER)
        S.running.copy_(S.active)
        if S.fcg_Pe is not None:
            if eta0 is None:
                S.fcg_Pe.zero_()
            else:                                   # (constrained_trust_regions.py:512-516: <grad c_k, eta0>, not zero)
                S.fcg_Pe.copy_(torch.stack([man.inner(S.x, gci, S.eta) for gci in S.gc], dim=1))

    def _tcg_step(self, problem, S, neq, Delta_cons, check_residual):
        """One truncated-CG iteration for every restart (robust_trust_regions.py:476-568, constrained_trust_regions.py:530-732)."""
        man = problem.manifold
        inner = man.inner
        x, g, delta = S.x, S.g, S.delta
        constrained = S.fc is not None
        Delta2 = S.Delta * S.Delta
        running = S.running.clone()
        stop = S.stop
        Hdelta = problem.hess(x, delta, grad_x=g)
        d_Hd = inner(x, delta, Hdelta)
        nz = d_Hd != 0
        alpha = torch.where(nz, S.z_r / torch.where(nz, d_Hd, torch.ones_like(d_Hd)), torch.zeros_like(d_Hd))
        e_Pe_new = torch.where(nz, S.e_Pe + 2 * alpha * S.e_Pd + alpha * alpha * S.d_Pd, S.e_Pe)
        if constrained:
            fc, fcg_Pe = S.fc, S.fcg_Pe
            C = fc.shape[1]
            is_ineq = torch.arange(C, device=x.device) >= neq
            fcg_Pd = torch.stack([inner(x, gci, delta) for gci in S.gc], dim=1)

            def cons_step(step):
                """violation of the linearised constraints after `step` along delta, and the step that stops at Delta_cons"""
                term = fc + fcg_Pe + step[:, None] * fcg_Pd
                term = torch.where(is_ineq[None, :], torch.clamp(term, max=0.0), term)
                cin = (term * term).sum(1)
                m = ((~is_ineq[None, :]) | (term < 0)).to(fc.dtype)         # equality constraints + violated inequalities
                qa = (m * fcg_Pd * fcg_Pd).sum(1)
                qb = 2.0 * ((m * fc * fcg_Pd).sum(1) + (m * fcg_Pe * fcg_Pd).sum(1))
                qc = (m * fc * fc).sum(1) + 2.0 * (m * fc * fcg_Pe).sum(1) + (m * fcg_Pe * fcg_Pe).sum(1) - Delta_cons ** 2
                disc = qb * qb - 4.0 * qa * qc
                tau_ = torch.where(disc >= 0, (-qb + disc.clamp(min=0).sqrt()) / (2.0 * qa), torch.zeros_like(disc))
                return cin, tau_
        # ---- leave through the trust-region boundary / negative curvature
        out = running & ((d_Hd <= 0) | (e_Pe_new >= Delta2))
        tau = (-S.e_Pd + (S.e_Pd * S.e_Pd + S.d_Pd * (Delta2 - S.e_Pe)).sqrt()) / S.d_Pd
        reason = torch.where(d_Hd <= 0, torch.full_like(stop, NEGATIVE_CURVATURE), torch.full_like(stop, EXCEEDED_TR))
        if constrained:
            tau = torch.where(torch.isnan(tau), torch.zeros_like(tau), tau)
            cin, tau_c = cons_step(tau)
            hit = cin > Delta_cons ** 2
            tau = torch.where(hit, tau_c, tau)
            reason = torch.where((d_Hd > 0) & hit, torch.full_like(stop, REACHED_CONSTRAINTS), reason)
        eta = torch.where(_bm(out, S.eta), S.eta + _bm(tau, delta) * delta, S.eta)
        Heta = torch.where(_bm(out, S.Heta), S.Heta + _bm(tau, Hdelta) * Hdelta, S.Heta)
        stop = torch.where(out, reason, stop)
        running = running & ~out
        # ---- leave because the linearised constraints are reached inside the trust region
        if constrained:
            cin, tau_c = cons_step(alpha)
            out = running & (cin > Delta_cons ** 2)
            eta = torch.where(_bm(out, eta), eta + _bm(tau_c, delta) * delta, eta)
            Heta = torch.where(_bm(out, Heta), Heta + _bm(tau_c, Hdelta) * Hdelta, Heta)
            stop = torch.where(out, torch.full_like(stop, REACHED_CONSTRAINTS), stop)
            running = running & ~out
        # ---- tentative step; reject it if the model did not decrease
        new_eta = eta + _bm(alpha, delta) * delta
        new_Heta = Heta + _bm(alpha, Hdelta) * Hdelta
        new_model = inner(x, new_eta, g) + 0.5 * inner(x, new_eta, new_Heta)
        out = running & ~(new_model < S.model_value)
        stop = torch.where(out, torch.full_like(stop, MODEL_INCREASED), stop)
        running = running & ~out
        eta = torch.where(_bm(running, eta), new_eta, eta)
        Heta = torch.where(_bm(running, Heta), new_Heta, Heta)
        S.model_value.copy_(torch.where(running, new_model, S.model_value))
        S.e_Pe.copy_(torch.where(running, e_Pe_new, S.e_Pe))
        r = torch.where(_bm(running, S.r), S.r + _bm(alpha, Hdelta) * Hdelta, S.r)
        norm_r = inner(x, r, r).clamp(min=0).sqrt()
        # ---- residual small enough
        if check_residual:
            target = S.norm_r0 * torch.minimum(S.norm_r0 ** self.theta, torch.full_like(S.norm_r0, self.kappa))
            out = running & (norm_r <= target)
            reason = torch.where(self.kappa < S.norm_r0 ** self.theta, torch.full_like(stop, REACHED_TARGET_LINEAR),
                                 torch.full_like(stop, REACHED_TARGET_SUPERLINEAR))
            stop = torch.where(out, reason, stop)
            running = running & ~out
        # ---- next search direction (only the restarts still running move on)
        z = r if self.use_rand else problem.precon(x, r)
        z_r_new = inner(x, z, r)
        beta = z_r_new / S.z_r
        new_delta = torch.where(_bm(running, delta), -z + _bm(beta, delta) * delta, delta)
        new_e_Pd = torch.where(running, beta * (S.e_Pd + alpha * S.d_Pd), S.e_Pd)
        new_d_Pd = torch.where(running, z_r_new + beta * beta * S.d_Pd, S.d_Pd)
        new_z_r = torch.where(running, z_r_new, S.z_r)
        if constrained:
            S.fcg_Pe.copy_(torch.where(running[:, None], fcg_Pe + alpha[:, None] * fcg_Pd, fcg_Pe))
        S.eta.copy_(eta)
        S.Heta.copy_(Heta)
        S.r.copy_(r)
        S.delta.copy_(new_delta)
        S.e_Pd.copy_(new_e_Pd)
        S.d_Pd.copy_(new_d_Pd)
        S.z_r.copy_(new_z_r)
        S.stop.copy_(stop)
        S.running.copy_(running)
        S.any_running.copy_(running.any())

    def _tcg_device(self, problem, x, g, Delta, active, mininner, maxinner, fc, gc, neq, Delta_cons, eta0=None):
        """The tCG loop on the device-resident state machine (csrc/spd_tcg.hip): per inner iteration one FD-point launch, the
        fused acquisition-gradient chain and one step launch; with hipGraphs the three are one replay."""
        from .. import ops
        ncons = 0 if fc is None else fc.shape[1]
        R, d = x.shape[0], x.shape[-1]
        key = ("dev", R, d, ncons, x.device)
        cache = problem.__dict__.setdefault("_tcg_cache", {})
        ent = cache.get(key)
        if ent is None:
            ent = {"T": ops.SpdTcg(R, d, ncons, x.device), "graph": None}
            cache[key] = ent
        T = ent["T"]
        fused = problem.fused
        args = (neq, Delta_cons, self.theta, self.kappa, mininner)

        def one_step():
            T.step(fused.egrad_mandel(T.fd_point()), *args)

        def begin():
            T.begin(x, g, torch.stack(gc) if ncons else None, fc, active, Delta)
            if eta0 is not None:                                    # use_rand (robust_trust_regions.py:411-415)
                T.begin_rand(eta0, heta0)

        heta0 = None if eta0 is None else problem.hess(x, eta0, grad_x=g)
        begin()
        graphs = bool(getattr(problem, "use_hip_graphs", False))
        if graphs and ent["graph"] is None:
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                one_step()                                          # warm-up outside capture
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                one_step()
            ent["graph"] = graph
            begin()                                                 # the warm-up and capture advanced the state
        for _ in range(int(maxinner)):
            if graphs:
                ent["graph"].replay()
            else:
                one_step()
            problem.n_grad += 1
            if not bool(T.any_running.item()):
                break
        return T.end()

    def _tcg(self, problem, x, g, Delta, active, mininner, maxinner, fc, gc, neq, Delta_cons, eta0=None):
        ncons = 0 if fc is None else fc.shape[1]
        if (eta0 is None or getattr(getattr(problem, "fused", None), "family", None) == "spd") and self._device_tcg_applies(problem, x, ncons):
            return self._tcg_device(problem, x, g, Delta, active, mininner, maxinner, fc, gc, neq, Delta_cons, eta0=eta0)
        graphs = bool(getattr(problem, "use_hip_graphs", False)) and x.is_cuda and eta0 is None
        key = (tuple(x.shape), x.device, ncons, neq, float(Delta_cons), int(mininner))
        cache = problem.__dict__.setdefault("_tcg_cache", {})
        ent = cache.get(key)
        if ent is None:
            ent = {"S": self._TcgState(x, ncons), "graphs": None}
            cache[key] = ent
        S = ent["S"]
        S.x.copy_(x)
        S.g.copy_(g)
        S.Delta.copy_(Delta)
        S.active.copy_(active)
        if ncons:
            S.fc.copy_(fc)
            for dst, src in zip(S.gc, gc):
                dst.copy_(src)
        if graphs and ent["graphs"] is None:
            # capture begin / first step (j < mininner: no residual test) / later steps once per problem and shape
            problem._capturing = True
            try:
                side = torch.cuda.Stream(device=x.device)
                side.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(side):
                    self._tcg_begin(problem, S)
                    self._tcg_step(problem, S, neq, Delta_cons, False)
                    self._tcg_step(problem, S, neq, Delta_cons, True)
                torch.cuda.current_stream(x.device).wait_stream(side)
                gb, g0, g1 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb):
                    self._tcg_begin(problem, S)
                with torch.cuda.graph(g0, pool=gb.pool()):
                    self._tcg_step(problem, S, neq, Delta_cons, False)
                with torch.cuda.graph(g1, pool=gb.pool()):
                    self._tcg_step(problem, S, neq, Delta_cons, True)
                ent["graphs"] = (gb, g0, g1)
            finally:
                problem._capturing = False
            S.x.copy_(x)
            S.g.copy_(g)
            S.Delta.copy_(Delta)
            S.active.copy_(active)
        if graphs:
            gb, g0, g1 = ent["graphs"]
            gb.replay()
        else:
            self._tcg_begin(problem, S, eta0)
        for j in range(int(maxinner)):
            check = j >= mininner
            if graphs:
                (g1 if check else g0).replay()
            else:
                self._tcg_step(problem, S, neq, Delta_cons, check)
            if not bool(S.any_running):
                break
        return S.eta.clone(), S.Heta.clone(), S.stop.clone()
