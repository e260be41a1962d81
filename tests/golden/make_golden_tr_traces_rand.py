#!/usr/bin/env python3
"""Per-iteration traces of the reference's TrustRegions with `use_rand=True` (robust_trust_regions.py:173-219: tCG started from a tiny random
tangent vector, no preconditioner; :196-219: the result compared with the Cauchy point) - development container only; needs /root/reference.
The random start is whatever `manifold.randvec` returns (here: a normal draw projected to the tangent space and normalised, the pymanopt
formula [3P], from numpy's seeded global stream); what makes the record reproducible by another implementation is that the vector HANDED TO
tCG is stored with every iteration (`eta_in`), so a test can replay it instead of drawing.  Same costs as tr_traces.npz (read from it), float64.

  * sph3, sph5: exact Hessian;  sph3 also with get_hessianfd;  spd3: get_hessianfd (the reference's SPD setting)
  * ConstrainedTrustRegions(use_rand=True) (constrained_trust_regions.py:207-262, 512-516: the linearised constraints start from
    <grad c, eta0>): sph3 with x[0] - 0.3 >= 0 from the starts of tr_traces.npz's constrained runs
-> tests/golden/tr_traces_rand.npz
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
MAXIT = T.MAXIT


def sphere_randvec(x):
    h = np.random.randn(*x.shape)
    p = h - np.dot(x, h) * x
    return p / np.linalg.norm(p)


def spd_randvec(x, man=None):
    h = np.random.randn(*x.shape)
    u = 0.5 * (h + h.T)
    return u / man.norm(x, u)


def traced_rand(cls):
    name = "_constrained_truncated_conjugate_gradient" if hasattr(cls, "_constrained_truncated_conjugate_gradient") \
        else "_truncated_conjugate_gradient"
    inner = getattr(cls, name)

    def wrapper(self, problem, x, fgradx, eta, Delta, *rest):
        eta_in = np.array(eta, dtype=np.float64, copy=True)
        out = inner(self, problem, x, fgradx, eta, Delta, *rest)
        self.trace.append((np.array(x, dtype=np.float64, copy=True), float(Delta), int(out[3]), int(out[2]), np.array(out[0], copy=True), eta_in))
        return out
    return type("TracedRand" + cls.__name__, (cls,), {name: wrapper})


def main():
    g = np.load(os.path.join(HERE, "tr_traces.npz"))
    out = {}
    torch.set_default_dtype(torch.float64)
    np.random.seed(20250930)
    for name, kind, fd, kw in (("sph3", "sphere", False, {}), ("sph5", "sphere", False, {}), ("sph3", "sphere", True, {}),
                               ("spd3", "spd", True, {"mingradnorm": 1e-4, "maxiter": MAXIT})):
        n = int(name[3:])
        Yt, wt, beta = torch.tensor(g[f"{name}_Y"]), torch.tensor(g[f"{name}_w"]), float(g[f"{name}_beta"])
        if kind == "sphere":
            man = base.SphereMan(n)
            man.randvec = sphere_randvec

            def cost(x, Yt=Yt, wt=wt, beta=beta):
                dd = base.sphere_distance_torch(x[None].double(), Yt)
                return -(wt * torch.exp(-beta * dd * dd)).sum()
            shape = (n,)
        else:
            man = base.SpdMan(n)
            man.randvec = types.MethodType(lambda self, x: spd_randvec(x, self), man)

            def cost(x, Yt=Yt, wt=wt, beta=beta):
                dist = base.affine_invariant_distance_torch(x[None].double(), Yt)
                return -(wt * torch.exp(-beta * dist * dist)).sum()
            shape = (n, n)
        x0 = g[f"{name}_x0"]
        rname = "rand_fd" if fd else "rand_exact"
        traces, finals, costs = [], [], []
        for xs in x0:
            prob = base.Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
            if fd:
                prob._hess = types.MethodType(base.get_hessianfd, prob)
            solver = traced_rand(base.TrustRegions)(use_rand=True, **kw)
            solver.trace = []
            x = solver.solve(prob, x=xs.copy())
            traces.append(solver.trace); finals.append(np.asarray(x)); costs.append(prob.cost(x))
        packed = T.pack([[t[:5] for t in tr] for tr in traces], finals, shape)
        S = len(traces)
        eta_in = np.full((S, MAXIT) + shape, np.nan)
        for s, tr in enumerate(traces):
            for k, t in enumerate(tr[:MAXIT]):
                eta_in[s, k] = t[5]
        packed["eta_in"], packed["x"], packed["f"], packed["ok"] = eta_in, np.stack(finals), np.array(costs), np.ones(S, dtype=bool)
        for k, v in packed.items():
            out[f"{name}_{rname}_f64_{k}"] = v
        print(name, rname, "iterations", packed["nit"], "f", np.array(costs).round(6), "|eta_in|", [float(np.linalg.norm(tr[0][5])) for tr in traces][:3], flush=True)
    # ---- constrained runs (drawn after all the unconstrained ones: their records above do not depend on these)
    from BoManifolds.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch
    # (spd3 with the lambda_max bound is not recorded: a restart on the bound shrinks its radius below the 1e-6 of the random start and the
    # reference's `while man.norm(x, eta) > Delta: eta = np.sqrt(np.sqrt(np.spacing(1)))` (:214-215) replaces the tangent vector by a SCALAR -
    # its next `man.norm(x, eta)` raises; this package scales the vector by that factor instead)
    for name, kind in (("sph3", "sphere"),):
        n = int(name[3:])
        Yt, wt, beta = torch.tensor(g[f"{name}_Y"]), torch.tensor(g[f"{name}_w"]), float(g[f"{name}_beta"])
        if kind == "sphere":
            man = base.SphereMan(n)
            man.randvec = sphere_randvec

            def cost(x, Yt=Yt, wt=wt, beta=beta):
                dd = base.sphere_distance_torch(x[None].double(), Yt)
                return -(wt * torch.exp(-beta * dd * dd)).sum()
            shape, x0, cons, fd, kw = (n,), g[f"{name}_con_x0"], [lambda x: x[0] - 0.3], False, {"mingradnorm": 1e-6, "maxiter": MAXIT}
        else:
            man = base.SpdMan(n)
            man.randvec = types.MethodType(lambda self, x: spd_randvec(x, self), man)

            def cost(x, Yt=Yt, wt=wt, beta=beta):
                dist = base.affine_invariant_distance_torch(x[None].double(), Yt)
                return -(wt * torch.exp(-beta * dist * dist)).sum()
            mx = float(g[f"{name}_maxeig"])
            shape, x0, cons, fd, kw = (n, n), g[f"{name}_x0"], [lambda x, m=mx: max_eigenvalue_constraint_torch(x, m)], True, {"mingradnorm": 1e-4, "maxiter": MAXIT}
        traces, finals, costs = [], [], []
        for xs in x0:
            prob = base.Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
            if fd:
                prob._hess = types.MethodType(base.get_hessianfd, prob)
            solver = traced_rand(base.ConstrainedTrustRegions)(use_rand=True, **kw)
            solver.trace = []
            x = solver.solve(prob, x=xs.copy(), ineq_constraints=cons)
            traces.append(solver.trace); finals.append(np.asarray(x)); costs.append(prob.cost(x))
        packed = T.pack([[t[:5] for t in tr] for tr in traces], finals, shape)
        S = len(traces)
        eta_in = np.full((S, MAXIT) + shape, np.nan)
        for s, tr in enumerate(traces):
            for k, t in enumerate(tr[:MAXIT]):
                eta_in[s, k] = t[5]
        packed["eta_in"], packed["x"], packed["f"], packed["ok"] = eta_in, np.stack(finals), np.array(costs), np.ones(S, dtype=bool)
        for k, v in packed.items():
            out[f"{name}_rand_con_f64_{k}"] = v
        print(name, "rand_con", "iterations", packed["nit"], "f", np.array(costs).round(6), flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "tr_traces_rand.npz"), **out)
    print("wrote tr_traces_rand.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB uncompressed")


if __name__ == "__main__":
    main()
