#!/usr/bin/env python3
"""Outer-iteration record of the reference's AugmentedLagrangeMethod with TrustRegions as the inner solver - the DEFAULT solver of
examples/bo_sphere/constrained_benchmark_examples/gabo_sphere_equality_constraints.py:95,200-203 and gabo_sphere_inequality_constraints.py:97,
238-241 (`AugmentedLagrangeMethod(maxiter=200, inner_solver=TrustRegions(maxiter=200), gammas_fact=0.05)`) - on the sphere
(development container only; needs /root/reference).  The solver classes are imported unmodified (stand-ins as in make_golden_tr.py, plus
`pymanopt.solvers.NelderMead` for an isinstance test and the sphere's `dist`); the point returned by every inner solve, the multipliers and
the penalty after every outer iteration are recorded by wrapping the inner solver's `solve`.  Costs: the kernel means of tr_traces.npz.

  * "eq":   the great circle x[1] = 0 of the equality example (:104-118), starts on it (its `sample_sphere_constrained`)
  * "ineq": the cap of half-angle pi/4 around e_0 of the inequality example (:108-121), starts inside it
-> tests/golden/alm.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_tr_traces as T  # noqa: E402
base = T.base


class _NelderMead:      # (only the target of an isinstance test, augmented_Lagrange_method.py:103,207)
    pass


sys.modules["pymanopt.solvers"].NelderMead = _NelderMead
sys.modules["pymanopt"].solvers = sys.modules["pymanopt.solvers"]
from BoManifolds.manifold_optimization.augmented_Lagrange_method import AugmentedLagrangeMethod  # noqa: E402

base.SphereMan.dist = staticmethod(lambda x, y: float(np.arccos(np.clip(np.dot(x, y), -1.0, 1.0))))    # [3P] pymanopt Sphere.dist
OUTER = 200


def main():
    g = np.load(os.path.join(HERE, "tr_traces.npz"))
    rng = np.random.default_rng(414)
    out = {}
    torch.set_default_dtype(torch.float64)
    for n in (3, 5):
        name = f"sph{n}"
        Y, w, beta = g[f"{name}_Y"], g[f"{name}_w"], float(g[f"{name}_beta"])
        Yt, wt = torch.tensor(Y, dtype=torch.float64), torch.tensor(w, dtype=torch.float64)
        man = base.SphereMan(n)

        def cost(x, Yt=Yt, wt=wt, beta=beta):
            dd = base.sphere_distance_torch(x[None].double(), Yt)
            return -(wt * torch.exp(-beta * dd * dd)).sum()
        # starts
        x_eq = rng.standard_normal((4, n))
        idx = np.arange(n) != 1
        x_eq[:, 1] = 0.0
        x_eq[:, idx] /= np.linalg.norm(x_eq[:, idx], axis=1, keepdims=True)
        ang = np.pi / 4
        x_in = rng.uniform(size=(4, n))
        x_in[:, 1:] = 2 * np.sin(ang) * x_in[:, 1:] - np.sin(ang)
        x_in[:, 0] = (1 - np.cos(ang)) * x_in[:, 0] + np.cos(ang)
        x_in /= np.linalg.norm(x_in, axis=1, keepdims=True)
        center = np.zeros(n); center[0] = 1.0

        def domain_constraint(x, center=center, ang=ang):            # gabo_sphere_inequality_constraints.py:112-119
            c = torch.Tensor(center).type(x.dtype)
            ip = torch.mm(x[None], c[:, None])
            ip = torch.max(torch.min(ip, torch.ones(1, dtype=x.dtype)), -torch.ones(1, dtype=x.dtype))
            return ang - torch.acos(ip)[0, 0]
        runs = {"eq": (x_eq, dict(eq_constraints=[lambda x: x[1] - 0.0])), "ineq": (x_in, dict(ineq_constraints=[domain_constraint]))}
        for rname, (x0s, skw) in runs.items():
            out[f"{name}_{rname}_x0"] = x0s
            S = len(x0s)
            xs = np.full((S, OUTER, n), np.nan)
            mult = np.full((S, OUTER), np.nan)
            nit = np.zeros(S, dtype=np.int64)
            xf, ff = np.zeros((S, n)), np.zeros(S)
            for s, x0 in enumerate(x0s):
                prob = base.Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
                inner = base.TrustRegions(maxiter=200)
                rec = []
                real = inner.solve

                def solve(problem, x=None, real=real, rec=rec):
                    r = real(problem, x)
                    rec.append(np.array(r, copy=True))
                    return r
                inner.solve = solve
                solver = AugmentedLagrangeMethod(maxiter=OUTER, inner_solver=inner, gammas_fact=0.05)
                x = solver.solve(prob, x=x0.copy(), **skw)
                nit[s] = len(rec)
                xs[s, :len(rec)] = np.stack(rec)
                xf[s], ff[s] = x, prob.cost(x)
            out[f"{name}_{rname}_xs"], out[f"{name}_{rname}_nit"], out[f"{name}_{rname}_x"], out[f"{name}_{rname}_f"] = xs, nit, xf, ff
            print(name, rname, "outer iterations", nit, "f", ff, "x[:2]", xf[:, :2].round(6).tolist(), flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "alm.npz"), **out)
    print("wrote alm.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB uncompressed")


if __name__ == "__main__":
    main()
