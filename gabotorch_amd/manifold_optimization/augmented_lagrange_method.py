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


class _Subproblem:
    def __init__(self, problem, eqs, ineqs, lambdas, gammas, rho):
        self.manifold = problem.manifold
        self.problem, self.eqs, self.ineqs, self.lambdas, self.gammas, self.rho = problem, eqs, ineqs, lambdas, gammas, rho

    def cost(self, x):                                       # (:273-288)
        c = self.problem.cost(x)
        for k, con in enumerate(self.ineqs):
            c += self.rho / 2.0 * max(0.0, self.lambdas[k] / self.rho - con.cost(x)) ** 2
        for k, con in enumerate(self.eqs):
            c += self.rho / 2.0 * (self.gammas[k] / self.rho + con.cost(x)) ** 2
        return c

    def grad(self, x):                                       # (:290-317)
        g = [gi.copy() for gi in self.problem.grad(x)]
        for k, con in enumerate(self.ineqs):
            v = con.cost(x)
            if self.lambdas[k] / self.rho - v > 0:
                cg = con.grad(x)
                for i in range(len(g)):
                    g[i] += (v * self.rho - self.lambdas[k]) * cg[i]
        for k, con in enumerate(self.eqs):
            v = con.cost(x)
            cg = con.grad(x)
            for i in range(len(g)):
                g[i] += (v * self.rho + self.gammas[k]) * cg[i]
        return g


class AugmentedLagrangeMethod:
    def __init__(self, inner_solver, bound=20, rho_init=1, thetarho=0.3, tau=0.8, starting_tolgradnorm=1e-3, ending_tolgradnorm=1e-6,
                 lambdas_fact=1.0, gammas_fact=1.0, maxiter=1000, maxtime=1000, minstepsize=1e-10):
        self.inner_solver = inner_solver
        self._bound, self._rho_init, self._thetarho, self._tau = bound, rho_init, thetarho, tau
        self._starting_tolgradnorm, self._ending_tolgradnorm = starting_tolgradnorm, ending_tolgradnorm
        self._lambdas_fact, self._gammas_fact = lambdas_fact, gammas_fact
        self._maxiter, self._maxtime, self._minstepsize = maxiter, maxtime, minstepsize
        self.log = {}

    def solve(self, problem, x=None, eq_constraints=None, ineq_constraints=None, lambdas=None, gammas=None, rho=None):
        """problem: .manifold (a Product of host manifolds: points are lists), .cost(x) -> float, .grad(x) -> Riemannian gradient.
        Constraints: objects with .cost / .grad (see _Constraint).  Returns the final point."""
        man = problem.manifold
        eqs = list(eq_constraints or [])
        ineqs = list(ineq_constraints or [])
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
            xbest, _ = self.inner_solver.solve(sub, xbest)
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
        return xbest
