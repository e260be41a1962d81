#!/usr/bin/env python3
"""Golden optima of the reference's trust-region solvers (run in the development container only; needs /root/reference).

The reference solvers (BoManifolds/manifold_optimization/{robust,constrained}_trust_regions.py, approximate_hessian.py,
pymanopt_addons/problem.py + PytorchBackend) are imported UNMODIFIED.  Two things they need are absent here and are supplied
as stand-ins that are NOT part of the reference tree:
  * `pymanopt.solvers.solver.Solver` (base class: constructor kwargs + _check_stopping_criterion; SURVEY App. B [3P]);
  * a manifold object: the duck type {inner, norm, retr, transp, egrad2rgrad, ehess2rhess, zerovec, dim, typicaldist}
    implemented with numpy from the formulas of SURVEY App. B (for SPD: exp/log are the reference's own vendored
    multiexp/multilog, pymanopt_addons/tools/multi.py).
Costs are kernel-mean functions built from the reference's own distance functions.
"""
import collections
import os
import sys
import time
import types

import numpy as np
import torch

REF = os.environ.get("GABO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
_R = collections.namedtuple("symeig", ["eigenvalues", "eigenvectors"])
torch.symeig = lambda A, eigenvectors=False, upper=True: _R(*torch.linalg.eigh(A, UPLO="U" if upper else "L"))
sys.path.insert(0, REF)


class Solver:   # stand-in for pymanopt.solvers.solver.Solver (0.2.x) [3P]
    def __init__(self, maxtime=1000, maxiter=1000, mingradnorm=1e-6, minstepsize=1e-10, maxcostevals=5000, logverbosity=0):
        self._maxtime, self._maxiter, self._mingradnorm = maxtime, maxiter, mingradnorm
        self._minstepsize, self._maxcostevals, self._logverbosity = minstepsize, maxcostevals, logverbosity
        self._optlog = None

    def _check_stopping_criterion(self, time0, iter=-1, gradnorm=float("inf"), stepsize=float("inf"), costevals=-1):
        reason = None
        if time.time() >= time0 + self._maxtime:
            reason = "time"
        elif iter >= self._maxiter:
            reason = "iter"
        elif gradnorm < self._mingradnorm:
            reason = "gradnorm"
        elif stepsize < self._minstepsize:
            reason = "stepsize"
        elif costevals >= self._maxcostevals:
            reason = "costevals"
        return reason

    def _start_optlog(self, *a, **k):
        pass

    def _stop_optlog(self, *a, **k):
        pass


for name in ("pymanopt", "pymanopt.solvers", "pymanopt.solvers.solver"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["pymanopt.solvers.solver"].Solver = Solver
sys.modules["pymanopt.solvers"].solver = sys.modules["pymanopt.solvers.solver"]
sys.modules["pymanopt"].solvers = sys.modules["pymanopt.solvers"]

from BoManifolds.manifold_optimization.robust_trust_regions import TrustRegions  # noqa: E402
from BoManifolds.manifold_optimization.constrained_trust_regions import ConstrainedTrustRegions, StrictConstrainedTrustRegions  # noqa: E402
from BoManifolds.manifold_optimization.approximate_hessian import get_hessianfd  # noqa: E402
from BoManifolds.pymanopt_addons.problem import Problem  # noqa: E402
from BoManifolds.pymanopt_addons.tools.multi import multiexp, multilog, multiprod, multisym, multitransp  # noqa: E402
from BoManifolds.Riemannian_utils.sphere_utils_torch import sphere_distance_torch  # noqa: E402
from BoManifolds.Riemannian_utils.spd_utils_torch import affine_invariant_distance_torch  # noqa: E402
from BoManifolds.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch  # noqa: E402


class SphereMan:
    def __init__(self, n):
        self._shape, self.dim, self.typicaldist = (n,), n - 1, np.pi
    inner = staticmethod(lambda x, u, v: float(np.tensordot(u, v, axes=u.ndim)))
    norm = staticmethod(lambda x, u: float(np.linalg.norm(u)))
    proj = staticmethod(lambda x, h: h - np.tensordot(x, h, axes=h.ndim) * x)
    egrad2rgrad = proj
    zerovec = staticmethod(lambda x: np.zeros_like(x))

    def ehess2rhess(self, x, eg, eh, u):
        return self.proj(x, eh) - np.tensordot(x, eg, axes=x.ndim) * u

    @staticmethod
    def retr(x, u):
        y = x + u
        return y / np.linalg.norm(y)

    def transp(self, x1, x2, d):
        return self.proj(x2, d)


class SpdMan:
    def __init__(self, n):
        self._n, self.dim, self.typicaldist = n, n * (n + 1) // 2, np.sqrt(n * (n + 1) / 2)

    @staticmethod
    def inner(x, u, v):
        return float(np.tensordot(np.linalg.solve(x, u), np.linalg.solve(x, v).T, axes=2))

    def norm(self, x, u):
        return float(np.sqrt(max(self.inner(x, u, u), 0.0)))
    zerovec = staticmethod(lambda x: np.zeros_like(x))
    egrad2rgrad = staticmethod(lambda x, g: x @ (0.5 * (g + g.T)) @ x)

    @staticmethod
    def ehess2rhess(x, eg, eh, u):
        s = lambda a: 0.5 * (a + a.T)   # noqa: E731
        return x @ s(eh) @ x + s(u @ s(eg) @ x)

    @staticmethod
    def retr(x, u):        # [3P] retr = exp = L multiexp(L^-1 U L^-T) L^T
        L = np.linalg.cholesky(x)
        Li = np.linalg.inv(L)
        return L @ multiexp(multisym(Li @ u @ Li.T), sym=True) @ L.T
    transp = staticmethod(lambda x1, x2, d: d)


def run(solver, problem, x0s, **kw):
    outs = []
    for x0 in x0s:
        outs.append(solver.solve(problem, x=x0.copy(), **kw))
    xs = np.stack(outs)
    fs = np.array([problem.cost(x) for x in outs])
    return xs, fs


def main():
    out = {}
    rng = np.random.default_rng(5)
    # ---------------- sphere S^2 and S^4: exact Hessian (stock TrustRegions) and FD Hessian
    for n in (3, 5):
        Y = rng.standard_normal((12, n)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        w = rng.uniform(0.2, 1.0, 12) * np.sign(rng.standard_normal(12))
        beta = 2.0
        Yt, wt = torch.tensor(Y), torch.tensor(w)

        def cost(x, Yt=Yt, wt=wt, beta=beta):
            d = sphere_distance_torch(x[None].double(), Yt)
            return -(wt * torch.exp(-beta * d * d)).sum()
        x0 = rng.standard_normal((6, n)); x0 /= np.linalg.norm(x0, axis=1, keepdims=True)
        man = SphereMan(n)
        prob = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        xs, fs = run(TrustRegions(), prob, x0)
        out[f"sph{n}_Y"], out[f"sph{n}_w"], out[f"sph{n}_beta"], out[f"sph{n}_x0"] = Y, w, np.float64(beta), x0
        out[f"sph{n}_exact_x"], out[f"sph{n}_exact_f"] = xs, fs
        prob2 = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        prob2._hess = types.MethodType(get_hessianfd, prob2)
        xs, fs = run(TrustRegions(), prob2, x0)
        out[f"sph{n}_fd_x"], out[f"sph{n}_fd_f"] = xs, fs
        # bound constraint x[0] >= 0.3 handled by the constrained solver (examples/gabo_sphere_bound_constraints.py style)
        prob3 = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        x0c = x0.copy(); x0c[:, 0] = np.abs(x0c[:, 0]) + 0.5; x0c /= np.linalg.norm(x0c, axis=1, keepdims=True)
        xs, fs = run(ConstrainedTrustRegions(mingradnorm=1e-6, maxiter=200), prob3, x0c, ineq_constraints=[lambda x: x[0] - 0.3])
        out[f"sph{n}_con_x0"], out[f"sph{n}_con_x"], out[f"sph{n}_con_f"] = x0c, xs, fs
        prob4 = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        xs, fs = run(StrictConstrainedTrustRegions(mingradnorm=1e-6, maxiter=200), prob4, x0c, ineq_constraints=[lambda x: x[0] - 0.3])
        out[f"sph{n}_strict_x"], out[f"sph{n}_strict_f"] = xs, fs
    # ---------------- SPD d=2,3: FD Hessian, unconstrained + max-eigenvalue constraint (examples/gabo_spd.py:136-138,183,200-203)
    for d in (2, 3):
        def rs(k, lo=0.3, hi=3.0):
            q = np.linalg.qr(rng.standard_normal((k, d, d)))[0]
            m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (k, d)), q)
            return 0.5 * (m + m.transpose(0, 2, 1))
        Y = rs(8)
        w = rng.uniform(0.2, 1.0, 8) * np.sign(rng.standard_normal(8))
        beta = 0.7
        Yt, wt = torch.tensor(Y), torch.tensor(w)

        def cost(x, Yt=Yt, wt=wt, beta=beta):
            dist = affine_invariant_distance_torch(x[None].double(), Yt)
            return -(wt * torch.exp(-beta * dist * dist)).sum()
        x0 = rs(5)
        man = SpdMan(d)
        prob = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        prob._hess = types.MethodType(get_hessianfd, prob)
        xs, fs = run(TrustRegions(mingradnorm=1e-4, maxiter=100), prob, x0)
        out[f"spd{d}_Y"], out[f"spd{d}_w"], out[f"spd{d}_beta"], out[f"spd{d}_x0"] = Y, w, np.float64(beta), x0
        out[f"spd{d}_fd_x"], out[f"spd{d}_fd_f"] = xs, fs
        prob2 = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        prob2._hess = types.MethodType(get_hessianfd, prob2)
        maxeig = 2.5
        x0c = rs(5, 0.3, 2.0)
        xs, fs = run(ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100), prob2, x0c,
                     ineq_constraints=[lambda x: max_eigenvalue_constraint_torch(x, maxeig)])
        out[f"spd{d}_con_x0"], out[f"spd{d}_con_x"], out[f"spd{d}_con_f"] = x0c, xs, fs
        prob3 = Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
        prob3._hess = types.MethodType(get_hessianfd, prob3)
        xs, fs = run(StrictConstrainedTrustRegions(mingradnorm=2e-4, maxiter=100, minstepsize=1e-4), prob3, x0c,
                     ineq_constraints=[lambda x: max_eigenvalue_constraint_torch(x, maxeig)])     # examples/hd_gabo_spd.py:194
        out[f"spd{d}_strict_x"], out[f"spd{d}_strict_f"] = xs, fs
        out[f"spd{d}_maxeig"] = np.float64(maxeig)
    np.savez_compressed(os.path.join(HERE, "trust_regions.npz"), **out)
    for k in sorted(out):
        if k.endswith("_f"):
            print(k, out[k])


if __name__ == "__main__":
    main()
