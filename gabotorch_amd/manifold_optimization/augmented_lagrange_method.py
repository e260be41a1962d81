"""Augmented Lagrangian method on manifolds (Liu & Boumal 2019) with the reference's parameterisation and update rules
(BoManifolds/manifold_optimization/augmented_Lagrange_method.py:29-326): bounded multipliers, penalty rho divided by `thetarho`
when the constraint violation did not shrink by `tau`, geometric tightening of the inner tolerance from `starting_tolgradnorm`
to `ending_tolgradnorm` over `maxiter` outer iterations.  Constraints: callables of the point, equalities satisfied at 0,
inequalities at >= 0.  Host-side (it moves the few reconstruction parameters of the nested-SPD mapping); the costs it is given
evaluate through the HIP kernels."""
import time

import numpy as np


class _Constraint:
    """cost / Riemannian gradient of one constraint function"""

    def __init__(self, manifold, value_and_egrad):
        self.manifold, self._vg = manifold, value_and_egrad

    def cost(self, x):
        return self._vg(x)[0]

    def grad(self, x):
        return self.manifold.egrad2rgrad(x, self._vg(x)[1])


def _as_constraint_problems(manifold, constraints, verbosity):
    """None / one constraint / a list of them -> a list of objects with .cost(x) and .grad(x).  A bare callable on torch tensors is
    wrapped the way the reference wraps it, `Problem(manifold, g, arg=torch.Tensor())` (augmented_Lagrange_method.py:133-136)."""
    import torch

    from ..pymanopt_addons.problem import Problem
    if constraints is None:
        return []
    if not isinstance(constraints, (list, tuple)):
        constraints = [constraints]
    return [c if hasattr(c, "cost") and hasattr(c, "grad") else Problem(manifold, c, arg=torch.Tensor(), verbosity=verbosity)
            for c in constraints]


def _axpy(g, a, cg):
    """g + a cg for a gradient that is an array or, on product manifolds, a list of arrays"""
    if isinstance(g, (list, tuple)):
        return [gi + a * ci for gi, ci in zip(g, cg)]
    return g + a * cg


class _Subproblem:
    """The unconstrained subproblem (:226-326) with pymanopt's problem attributes: cost, grad, the finite-difference hess the reference
    binds (:324 `get_hessianfd` - of the ORIGINAL problem, see hess below), precon, verbosity."""

    def __init__(self, problem, eqs, ineqs, lambdas, gammas, rho):
        self.manifold = problem.manifold
        self.problem, self.eqs, self.ineqs, self.lambdas, self.gammas, self.rho = problem, eqs, ineqs, lambdas, gammas, rho
        self.verbosity = getattr(problem, "verbosity", 0)
        self.precon = getattr(problem, "precon", None) or (lambda x, d: d)

    def hess(self, x, a):
        # The reference binds `types.MethodType(get_hessianfd, problem)` (:324) - to the ORIGINAL problem, not to the subproblem it has just
        # built: the finite differences are those of the objective's gradient alone, the penalty terms contribute nothing to the model's
        # curvature.  Restated as written (tests/golden/alm.npz: with it every outer iterate of the reference's runs is reproduced to
        # 1e-8; with the subproblem's own gradient - rounds 2-4 - the first inner solve already lands 1e-3 away and single starts end in
        # another local optimum).  Inner solvers without a Hessian (the reconstruction's conjugate gradients) never call this.
        from .approximate_hessian import get_hessianfd
        return get_hessianfd(self.problem, x, a)

    @property
    def prefetch(self):
        """the wrapped problem's batched look-ahead (see LineSearchAdaptive.search), when it has one"""
        return getattr(self.problem, "prefetch", None)

    def cost(self, x):                                       # (:273-288)
        c = self.problem.cost(x)
        for k, con in enumerate(self.ineqs):
            c += self.rho / 2.0 * max(0.0, self.lambdas[k] / self.rho - con.cost(x)) ** 2
        for k, con in enumerate(self.eqs):
            c += self.rho / 2.0 * (self.gammas[k] / self.rho + con.cost(x)) ** 2
        return c

    def grad(self, x):                                       # (:290-317)
        g = self.problem.grad(x)
        g = [np.array(gi, dtype=float) for gi in g] if isinstance(g, (list, tuple)) else np.array(g, dtype=float)
        for k, con in enumerate(self.ineqs):
            v = con.cost(x)
            if self.lambdas[k] / self.rho - v > 0:
                g = _axpy(g, v * self.rho - self.lambdas[k], con.grad(x))
        for k, con in enumerate(self.eqs):
            g = _axpy(g, con.cost(x) * self.rho + self.gammas[k], con.grad(x))
        return g


class AugmentedLagrangeMethod:
    def __init__(self, inner_solver, bound=20, rho_init=1, thetarho=0.3, tau=0.8, starting_tolgradnorm=1e-3, ending_tolgradnorm=1e-6,
                 lambdas_fact=1.0, gammas_fact=1.0, maxiter=1000, maxtime=1000, minstepsize=1e-10, logverbosity=0, **_solver_kwargs):
        self.inner_solver = inner_solver
        self._bound, self._rho_init, self._thetarho, self._tau = bound, rho_init, thetarho, tau
        self._starting_tolgradnorm, self._ending_tolgradnorm = starting_tolgradnorm, ending_tolgradnorm
        self._lambdas_fact, self._gammas_fact = lambdas_fact, gammas_fact
        self._maxiter, self._maxtime, self._minstepsize = maxiter, maxtime, minstepsize
        self._logverbosity = logverbosity
        self.log = {}

    def solve(self, problem, x=None, eq_constraints=None, ineq_constraints=None, lambdas=None, gammas=None, rho=None):
        """problem: .manifold, .cost(x) -> float, .grad(x) -> Riemannian gradient (points are arrays, or lists of arrays on a Product of
        manifolds).  Constraints, as in the reference (:72-136): one or a list of callables on torch tensors (each wrapped in a
        `Problem(manifold, g, arg=torch.Tensor())`) - or objects that already have .cost / .grad (see _Constraint).  The inner solver may
        return the point or (point, log).  Returns the final point, (point, log) when logverbosity >= 1 (:222-225)."""
        man = problem.manifold
        eqs = _as_constraint_problems(man, eq_constraints, getattr(problem, "verbosity", 0))
        ineqs = _as_constraint_problems(man, ineq_constraints, getattr(problem, "verbosity", 0))
        xbest = man.rand() if x is None else x
        xprev = xbest
        lambdas = self._lambdas_fact * np.ones(len(ineqs)) if lambdas is None else np.asarray(lambdas, dtype=float)
        gammas = self._gammas_fact * np.ones(len(eqs)) if gammas is None else np.asarray(gammas, dtype=float)
        rho = self._rho_init if rho is None else rho
        oldacc = np.inf
        tol = self._starting_tolgradnorm
        theta_tol = (self._ending_tolgradnorm / self._starting_tolgradnorm) ** (1.0 / self._maxiter)
        time0 = time.time()
        k = 0
        reason = "max iterations"
        while True:
            sub = _Subproblem(problem, eqs, ineqs, lambdas, gammas, rho)
            self.inner_solver._mingradnorm = tol
            res = self.inner_solver.solve(sub, xbest)
            xbest = res[0] if isinstance(res, tuple) and len(res) == 2 and isinstance(res[1], dict) else res
            newacc = 0.0
            # (:179-183) NOTE the reference raises an inequality multiplier by + rho g(x) although its constraints are satisfied at
            # g >= 0 (the usual update is max(lambda - rho g, 0)): restated as written; its own callers only pass equalities
            for c, con in enumerate(ineqs):
                v = con.cost(xbest)
                newacc = max(newacc, abs(max(-lambdas[c] / rho, v)))
                lambdas[c] = min(self._bound, max(lambdas[c] + rho * v, 0.0))
            for c, con in enumerate(eqs):                    # (:185-188)
                v = con.cost(xbest)
                newacc = max(newacc, abs(v))
                gammas[c] = min(self._bound, max(-self._bound, gammas[c] + rho * v))
            if k == 0 or newacc > self._tau * oldacc:        # (:191-193)
                rho = rho / self._thetarho
            oldacc = newacc
            tol = max(self._ending_tolgradnorm, tol * theta_tol)
            k += 1
            step = man.dist(xbest, xprev)
            if time.time() - time0 >= self._maxtime:
                reason = "max time"
            elif k >= self._maxiter:
                reason = "max iterations"
            elif step < self._minstepsize:
                reason = "min step size"
            elif tol <= self._ending_tolgradnorm:
                reason = "min grad norm"
            else:
                xprev = xbest
                continue
            break
        self.log = {"iterations": k, "stop_reason": reason, "violation": oldacc, "rho": rho, "time": time.time() - time0,
                    "lambdas": lambdas, "gammas": gammas}
        return (xbest, self.log) if self._logverbosity >= 1 else xbest


class _BatchedSubproblem:
    """The subproblem of every restart at once, for the lock-step trust regions: cost and gradient of the wrapped batched problem (the fused
    acquisition evaluation when it has one) plus the penalty terms (:273-317, autograd through the constraint callables), and - as the
    reference binds it (:324) - the finite-difference Hessian of the ORIGINAL problem.  lambdas: R x n_ineq, gammas: R x n_eq, rho: R."""
    approx_hessian = True
    fused = None
    use_hip_graphs = False

    def __init__(self, problem, eqs, ineqs, lambdas, gammas, rho):
        self.problem, self.eqs, self.ineqs, self.lambdas, self.gammas, self.rho = problem, eqs, ineqs, lambdas, gammas, rho
        self.manifold, self.precon = problem.manifold, problem.precon
        self.n_cost = self.n_grad = 0

    def _penalty(self, x, need_grad):
        """-> (penalty R, its Euclidean gradient or None).  The gradient is assembled per constraint as the reference does (:290-317) - an
        inequality contributes (g rho - lambda) grad g ONLY where lambda / rho - g > 0 - and not by differentiating the clamped square: where
        a constraint is inactive its gradient is never looked at, and it may be infinite there (the cap constraint angle - acos<x, c> at its
        own centre: 0 x inf inside autograd would be NaN)."""
        import torch

        from .batched_trust_regions import BatchedTrustRegions
        R = x.shape[0]
        bm = lambda m: m.reshape((R,) + (1,) * (x.dim() - 1))      # noqa: E731
        p = torch.zeros(R, dtype=x.dtype, device=x.device)
        g = torch.zeros_like(x) if need_grad else None

        def value_and_grad(con):
            xx = x.detach().clone().requires_grad_(need_grad)
            with torch.set_grad_enabled(need_grad):
                v = BatchedTrustRegions._call_constraint(con, xx).to(x.dtype)
                gv = None
                if need_grad:
                    (gv,) = torch.autograd.grad(v.sum(), xx, allow_unused=True)
                    gv = torch.zeros_like(x) if gv is None else gv
            return v.detach(), gv
        for k, con in enumerate(self.ineqs):           # (:279-282, :295-307: g >= 0 is satisfied)
            v, gv = value_and_grad(con)
            slack = self.lambdas[:, k] / self.rho - v
            act = slack > 0
            p = p + torch.where(act, self.rho / 2.0 * slack ** 2, torch.zeros_like(p))
            if need_grad:
                g = g + torch.where(bm(act), bm(v * self.rho - self.lambdas[:, k]) * gv, torch.zeros_like(gv))
        for k, con in enumerate(self.eqs):             # (:284-287, :309-317)
            v, gv = value_and_grad(con)
            p = p + self.rho / 2.0 * (self.gammas[:, k] / self.rho + v) ** 2
            if need_grad:
                g = g + bm(v * self.rho + self.gammas[:, k]) * gv
        return p, g

    def cost(self, x):
        self.n_cost += 1
        return self.problem.cost(x) + self._penalty(x, False)[0]

    def cost_egrad(self, x, create_graph=False):
        self.n_grad += 1
        f, eg, xx = self.problem.cost_egrad(x)
        # (kept for hess below: the ORIGINAL problem's Euclidean gradient at this very tensor, its Riemannian form made on first use)
        self._orig = [x, x._version, eg.detach(), None]
        p, g = self._penalty(x, True)
        return f + p, eg.detach() + g, xx

    def cost_grad(self, x):
        f, eg, _ = self.cost_egrad(x)
        return f, self.manifold.egrad2rgrad(x, eg)

    def grad(self, x):
        return self.cost_grad(x)[1]

    def hess(self, x, u, grad_x=None):
        # the original problem's own gradient differences (`grad_x` is the SUBPROBLEM's gradient).  Its gradient at x was evaluated by the
        # cost_egrad call that opened this trust-region iteration: handed over when x is that same tensor, unmodified - otherwise problem.hess
        # evaluates it again (one more gradient evaluation per Hessian-vector product, which is what every tCG step of every inner solve paid)
        o = getattr(self, "_orig", None)
        g0 = None
        if o is not None and o[0] is x and o[1] == x._version:
            if o[3] is None:
                o[3] = self.manifold.egrad2rgrad(x, o[2])
            g0 = o[3]
        return self.problem.hess(x, u, g0)


def _solve_batched(self, problem, x, eq_constraints=None, ineq_constraints=None):
    """The method on ALL restarts in lock step (R x point tensors, the inner solver one of this package's trust regions): the same updates
    per restart as `solve` - which is the reference's loop (:66-225) and stays the statement tests pin to the reference's record - with the
    inner solves batched.  A restart that has met its stopping criterion keeps its point while the others go on.  problem: BatchedProblem."""
    import torch
    man = problem.manifold
    eqs = [] if eq_constraints is None else (list(eq_constraints) if isinstance(eq_constraints, (list, tuple)) else [eq_constraints])
    ineqs = [] if ineq_constraints is None else (list(ineq_constraints) if isinstance(ineq_constraints, (list, tuple)) else [ineq_constraints])
    from .batched_trust_regions import BatchedTrustRegions
    R, dt, dev = x.shape[0], x.dtype, x.device
    full = lambda cols, v: torch.full((R, cols), float(v), dtype=dt, device=dev)      # noqa: E731
    lambdas, gammas = full(len(ineqs), self._lambdas_fact), full(len(eqs), self._gammas_fact)
    rho = torch.full((R,), float(self._rho_init), dtype=dt, device=dev)
    oldacc = torch.full((R,), float("inf"), dtype=dt, device=dev)
    tol = self._starting_tolgradnorm
    theta_tol = (self._ending_tolgradnorm / self._starting_tolgradnorm) ** (1.0 / self._maxiter)
    active = torch.ones(R, dtype=torch.bool, device=dev)
    xbest = x.detach().clone()
    xprev = xbest
    bm = lambda m: m.reshape((R,) + (1,) * (xbest.dim() - 1))      # noqa: E731
    time0 = time.time()
    k = 0
    outer = torch.zeros(R, dtype=torch.long, device=dev)
    while True:
        sub = _BatchedSubproblem(problem, eqs, ineqs, lambdas, gammas, rho)
        self.inner_solver._mingradnorm = tol
        xnew = self.inner_solver.solve(sub, xbest)
        xbest = torch.where(bm(active), xnew, xbest)
        newacc = torch.zeros(R, dtype=dt, device=dev)
        with torch.no_grad():
            for c, con in enumerate(ineqs):                     # (:179-183, restated as written: see `solve`)
                v = BatchedTrustRegions._call_constraint(con, xbest).to(dt)
                newacc = torch.maximum(newacc, torch.maximum(-lambdas[:, c] / rho, v).abs())
                lambdas[:, c] = torch.where(active, torch.clamp(lambdas[:, c] + rho * v, min=0.0, max=float(self._bound)), lambdas[:, c])
            for c, con in enumerate(eqs):                       # (:185-188)
                v = BatchedTrustRegions._call_constraint(con, xbest).to(dt)
                newacc = torch.maximum(newacc, v.abs())
                gammas[:, c] = torch.where(active, torch.clamp(gammas[:, c] + rho * v, min=-float(self._bound), max=float(self._bound)), gammas[:, c])
        grow = active & ((newacc > self._tau * oldacc) if k > 0 else torch.ones_like(active))       # (:191-193)
        rho = torch.where(grow, rho / self._thetarho, rho)
        oldacc = torch.where(active, newacc, oldacc)
        tol = max(self._ending_tolgradnorm, tol * theta_tol)
        k += 1
        outer += active.long()
        # pymanopt's dist on the sphere is arccos<x, y>; elsewhere the manifold's own, or the chordal distance
        if hasattr(man, "dist"):
            step = torch.as_tensor(man.dist(xbest, xprev), dtype=dt, device=dev).reshape(R)
        elif xbest.dim() == 2:
            step = torch.acos(torch.clamp((xbest * xprev).sum(-1), -1.0, 1.0))
        else:
            step = (xbest - xprev).flatten(1).norm(dim=1)
        active = active & ~(step < self._minstepsize)
        if (time.time() - time0 >= self._maxtime or k >= self._maxiter or tol <= self._ending_tolgradnorm or not bool(active.any())):
            break
        xprev = xbest
    self.log = {"iterations": k, "per_restart_iterations": outer, "violation": oldacc, "rho": rho, "time": time.time() - time0,
                "lambdas": lambdas, "gammas": gammas, "batched": True}
    return xbest


AugmentedLagrangeMethod.solve_batched = _solve_batched
