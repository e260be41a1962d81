"""Riemannian conjugate gradients with the adaptive backtracking line search - the default solver of the reference's surrogate fit
(`pyman_solvers.ConjugateGradient(maxiter=500)`, manifold_gp_fit.py:57) - for products of host-side manifolds.
[3P] pymanopt 0.2.x `ConjugateGradient` / `LineSearchAdaptive`, restated from memory (unpinned): Hestenes-Stiefel
beta clipped at 0, restart when the direction is not a descent direction, step shrunk by 1/2 until f <= f0 + 1/2 alpha df0
(at most 10 cost evaluations), next initial step = last accepted (x2 if it was accepted without exactly one contraction)."""
import time

import numpy as np


def _axpy(a, x, y):
    """a * x + y on (lists of) arrays"""
    if isinstance(x, list):
        return [a * xi + yi for xi, yi in zip(x, y)]
    return a * x + y


def _scale(a, x):
    return [a * xi for xi in x] if isinstance(x, list) else a * x


class LineSearchAdaptive:
    def __init__(self, contraction_factor=0.5, suff_decr=0.5, maxiter=10, initial_stepsize=1.0):
        self.contraction_factor, self.suff_decr, self.maxiter, self.initial_stepsize = contraction_factor, suff_decr, maxiter, initial_stepsize
        self._oldalpha = None

    def search(self, objective, man, x, d, f0, df0, prefetch=None):
        """prefetch (optional): callable taking a list of points whose objective values will be asked for next - a problem whose
        evaluations are device launches evaluates them together (the first trial step and the first contraction in one launch: about
        half of the searches contract once).  It changes which launches compute the values, never the values or the steps taken."""
        norm_d = man.norm(x, d)
        alpha = self._oldalpha if self._oldalpha is not None else self.initial_stepsize / norm_d
        alpha = float(alpha)
        nextx = None
        if prefetch is not None:
            steps = (alpha, alpha * self.contraction_factor)
            if hasattr(man, "retr_steps"):
                newx, nextx = man.retr_steps(x, d, steps)
            else:
                newx, nextx = (man.retr(x, _scale(t, d)) for t in steps)
            prefetch([newx, nextx])
        else:
            newx = man.retr(x, _scale(alpha, d))
        newf = objective(newx)
        evals = 1
        while newf > f0 + self.suff_decr * alpha * df0 and evals <= self.maxiter:
            alpha *= self.contraction_factor
            newx = nextx if nextx is not None else man.retr(x, _scale(alpha, d))
            nextx = None
            newf = objective(newx)
            evals += 1
        if newf > f0:
            alpha, newx, newf = 0.0, x, f0
        self._oldalpha = alpha if evals == 2 else 2.0 * alpha
        return alpha * norm_d, newx, newf


class ConjugateGradient:
    def __init__(self, maxiter=1000, maxtime=1000, mingradnorm=1e-6, minstepsize=1e-10, orth_value=np.inf, logverbosity=0):
        self.maxiter, self.maxtime, self.mingradnorm, self.minstepsize = maxiter, maxtime, mingradnorm, minstepsize
        self.orth_value = orth_value
        self._logverbosity = logverbosity

    @property
    def _mingradnorm(self):          # pymanopt's attribute name: the augmented Lagrangian method tightens it between subproblems
        return self.mingradnorm

    @_mingradnorm.setter
    def _mingradnorm(self, value):
        self.mingradnorm = float(value)

    def solve(self, problem, x=None):
        """problem: object with .manifold, .cost(x) -> float, .grad(x) -> Riemannian gradient.  Returns (x, log)."""
        man = problem.manifold
        linesearch = LineSearchAdaptive()
        if x is None:
            x = man.rand()
        time0 = time.time()
        cost = problem.cost(x)
        grad = problem.grad(x)
        gradnorm = man.norm(x, grad)
        grad_grad = man.inner(x, grad, grad)
        desc = _scale(-1.0, grad)
        stepsize, it, reason = np.nan, 0, "max iterations"
        history = [cost]
        while True:
            if gradnorm < self.mingradnorm:
                reason = "min grad norm"
                break
            if it >= self.maxiter:
                break
            if time.time() - time0 >= self.maxtime:
                reason = "max time"
                break
            if stepsize < self.minstepsize:
                reason = "min step size"
                break
            df0 = man.inner(x, grad, desc)
            if df0 >= 0:                              # not a descent direction: restart from steepest descent
                desc = _scale(-1.0, grad)
                df0 = -grad_grad
            stepsize, newx, newcost = linesearch.search(problem.cost, man, x, desc, cost, df0, prefetch=getattr(problem, "prefetch", None))
            newgrad = problem.grad(newx)
            newgradnorm = man.norm(newx, newgrad)
            new_gg = man.inner(newx, newgrad, newgrad)
            oldgrad = man.transp(x, newx, grad)
            orth = man.inner(newx, oldgrad, newgrad) / new_gg if new_gg > 0 else 0.0
            if abs(orth) >= self.orth_value:
                desc = _scale(-1.0, newgrad)
            else:
                desc = man.transp(x, newx, desc)
                diff = _axpy(-1.0, oldgrad, newgrad)
                den = man.inner(newx, diff, desc)
                beta = max(0.0, man.inner(newx, newgrad, diff) / den) if den != 0 else 1.0      # Hestenes-Stiefel
                desc = _axpy(beta, desc, _scale(-1.0, newgrad))
            x, cost, grad, gradnorm, grad_grad = newx, newcost, newgrad, newgradnorm, new_gg
            history.append(cost)
            it += 1
        return x, {"iterations": it, "stop_reason": reason, "final_cost": cost, "final_gradnorm": gradnorm, "cost_history": history,
                   "time": time.time() - time0}
