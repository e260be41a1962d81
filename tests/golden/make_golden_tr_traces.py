#!/usr/bin/env python3
"""Golden `get_hessianfd` values and per-iteration TRACES of the reference's trust-region solvers (development container only;
needs /root/reference).  Complements make_golden_tr.py (end optima) with what SURVEY 8c item 6 asks for:

  * `get_hessianfd(x, a)` (approximate_hessian.py:11-62) at fixed (x, a) on S^2, S^4, S^2_++, S^3_++, S^5_++;
  * for TrustRegions / ConstrainedTrustRegions / StrictConstrainedTrustRegions: the iterate x_k, the radius Delta_k, the tCG stop
    reason and inner-iteration count of EVERY outer iteration (recorded by wrapping the solvers' tCG method - the solver code itself is
    imported unmodified), plus the end point and cost;
  * a config-4-sized problem: S^5_++, 50 kernel-mean terms, lambda_max <= 5 constraint, 8 fixed starts.

Everything is produced twice: with the reference exactly as it is ("f32": `eig_values = torch.zeros(...)` in spd_utils_torch.py:108
makes the eigenvalues - and so every cost and gradient - single precision, and get_hessianfd divides gradient differences by
c = 2^-14 / |a|, amplifying that noise 16 000-fold) and with torch's DEFAULT dtype set to float64 while the same code runs ("f64":
the same statements, the buffer is then double).  The second is the arithmetic the HIP path implements; the distance between the two
is the reference's own noise floor and bounds what any fp64 implementation can reproduce of the first.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_tr as base  # noqa: E402  (installs the pymanopt Solver stand-in, imports the reference solvers)

from BoManifolds.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch  # noqa: E402

MAXIT = 100


def traced(cls):
    """subclass recording (x, Delta, stop reason, inner iterations) at every call of the solver's tCG routine"""
    name = "_constrained_truncated_conjugate_gradient" if hasattr(cls, "_constrained_truncated_conjugate_gradient") \
        else "_truncated_conjugate_gradient"
    inner = getattr(cls, name)

    def wrapper(self, problem, x, fgradx, eta, Delta, *rest):
        out = inner(self, problem, x, fgradx, eta, Delta, *rest)
        self.trace.append((np.array(x, dtype=np.float64, copy=True), float(Delta), int(out[3]), int(out[2]), np.array(out[0], copy=True)))
        return out
    return type("Traced" + cls.__name__, (cls,), {name: wrapper})


def pack(traces, finals, shape):
    S = len(traces)
    xs = np.full((S, MAXIT + 1) + shape, np.nan)
    eta = np.full((S, MAXIT) + shape, np.nan)
    delta = np.full((S, MAXIT), np.nan)
    stop = np.full((S, MAXIT), -1, dtype=np.int64)
    numit = np.full((S, MAXIT), -1, dtype=np.int64)
    nit = np.zeros(S, dtype=np.int64)
    for s, tr in enumerate(traces):
        nit[s] = len(tr)
        for k, (x, d, st, ni, e) in enumerate(tr[:MAXIT]):
            xs[s, k], delta[s, k], stop[s, k], numit[s, k], eta[s, k] = x, d, st, ni, e
        xs[s, min(len(tr), MAXIT)] = finals[s]
    return {"xs": xs, "eta": eta, "delta": delta, "stop": stop, "numit": numit, "nit": nit}


def solve_all(solver_cls, kwargs, make_problem, x0s, shape, **solve_kw):
    traces, finals, costs, ok = [], [], [], []
    for x0 in x0s:
        prob = make_problem()
        solver = traced(solver_cls)(**kwargs)
        solver.trace = []
        try:
            x = solver.solve(prob, x=x0.copy(), **solve_kw)
            traces.append(solver.trace)
            finals.append(np.asarray(x))
            costs.append(prob.cost(x))
            ok.append(True)
        except RuntimeError as err:                 # the reference itself fails from this start (e.g. torch.cholesky of a proposal)
            print("   reference solver raised:", str(err).splitlines()[0][:120], flush=True)
            traces.append(solver.trace)
            finals.append(np.full(shape, np.nan))
            costs.append(np.nan)
            ok.append(False)
    out = pack(traces, finals, shape)
    out["x"] = np.stack(finals)
    out["f"] = np.array(costs)
    out["ok"] = np.array(ok)
    return out


def rand_spd(rng, k, d, lo, hi):
    q = np.linalg.qr(rng.standard_normal((k, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (k, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def main():
    out = {}
    rng = np.random.default_rng(77)
    cases = []
    for n in (3, 5):
        Y = rng.standard_normal((12, n)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
        w = rng.uniform(0.2, 1.0, 12) * np.sign(rng.standard_normal(12))
        x0 = rng.standard_normal((4, n)); x0 /= np.linalg.norm(x0, axis=1, keepdims=True)
        cases.append(("sph%d" % n, "sphere", n, Y, w, 2.0, x0, None))
    for d, nterms, starts in ((2, 8, 4), (3, 8, 4), (5, 50, 8)):
        Y = rand_spd(rng, nterms, d, 0.3, 3.0) if d < 5 else rand_spd(rng, nterms, d, 1e-3 + 0.05, 5.0)
        w = rng.uniform(0.2, 1.0, nterms) * np.sign(rng.standard_normal(nterms))
        x0 = rand_spd(rng, starts, d, 0.3, 2.0) if d < 5 else rand_spd(rng, starts, d, 0.3, 4.5)
        cases.append(("spd%d" % d, "spd", d, Y, w, 0.7 if d < 5 else 0.25 + float(np.log(2.0)), x0, 2.5 if d < 5 else 5.0))
        if d == 5:
            # the same S^5_++ problem with the bound BELOW the unconstrained optimum's largest eigenvalue (3.31): the constraint is active
            cases.append(("spd5c", "spd", d, Y, w, 0.25 + float(np.log(2.0)), rand_spd(rng, starts, d, 0.3, 2.5), 2.8))

    for name, kind, n, Y, w, beta, x0, maxeig in cases:
        out[f"{name}_Y"], out[f"{name}_w"], out[f"{name}_beta"], out[f"{name}_x0"] = Y, w, np.float64(beta), x0
        if maxeig is not None:
            out[f"{name}_maxeig"] = np.float64(maxeig)
        shape = (n,) if kind == "sphere" else (n, n)
        man = base.SphereMan(n) if kind == "sphere" else base.SpdMan(n)
        # tangent vectors for the Hessian-vector fixture
        if kind == "sphere":
            a = rng.standard_normal(x0.shape)
            a = a - np.sum(a * x0, axis=1, keepdims=True) * x0
        else:
            a = rng.standard_normal(x0.shape)
            a = 0.5 * (a + a.transpose(0, 2, 1))
        a[1] *= 1e-3                                   # a short one: c = 2^-14 / |a| large
        out[f"{name}_hv_a"] = a
        for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
            torch.set_default_dtype(dtype)
            Yt, wt = torch.tensor(Y, dtype=torch.float64), torch.tensor(w, dtype=torch.float64)
            if kind == "sphere":
                def cost(x, Yt=Yt, wt=wt, beta=beta):
                    dd = base.sphere_distance_torch(x[None].double(), Yt)
                    return -(wt * torch.exp(-beta * dd * dd)).sum()
            else:
                def cost(x, Yt=Yt, wt=wt, beta=beta):
                    dist = base.affine_invariant_distance_torch(x[None].double(), Yt)
                    return -(wt * torch.exp(-beta * dist * dist)).sum()

            def make_problem(fd=True, cost=cost, man=man):
                p = base.Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
                if fd:
                    p._hess = types.MethodType(base.get_hessianfd, p)
                return p
            # ---- (i) get_hessianfd at fixed (x, a)
            p = make_problem()
            out[f"{name}_hv_cost_{tag}"] = np.array([p.cost(x) for x in x0])
            out[f"{name}_hv_grad_{tag}"] = np.stack([np.array(p.grad(x)) for x in x0])
            out[f"{name}_hv_fd_{tag}"] = np.stack([np.array(p.hess(x, ai)) for x, ai in zip(x0, a)])
            # ---- (ii) traces
            if kind == "sphere":
                runs = {"tr_fd": (base.TrustRegions, {}, {}, True),
                        "tr_exact": (base.TrustRegions, {}, {}, False)}
                x0c = x0.copy(); x0c[:, 0] = np.abs(x0c[:, 0]) + 0.5; x0c /= np.linalg.norm(x0c, axis=1, keepdims=True)
                out[f"{name}_con_x0"] = x0c
                con = [lambda x: x[0] - 0.3]
                cruns = {"con": (base.ConstrainedTrustRegions, {"mingradnorm": 1e-6, "maxiter": MAXIT}),
                         "strict": (base.StrictConstrainedTrustRegions, {"mingradnorm": 1e-6, "maxiter": MAXIT})}
            else:
                runs = {"tr_fd": (base.TrustRegions, {"mingradnorm": 1e-4, "maxiter": MAXIT}, {}, True)}
                x0c = x0
                con = [lambda x, m=maxeig: max_eigenvalue_constraint_torch(x, m)]
                cruns = {"con": (base.ConstrainedTrustRegions, {"mingradnorm": 1e-4, "maxiter": MAXIT}),
                         "strict": (base.StrictConstrainedTrustRegions, {"mingradnorm": 2e-4, "maxiter": MAXIT, "minstepsize": 1e-4})}
            for rname, (cls, kw, skw, fd) in runs.items():
                res = solve_all(cls, kw, lambda fd=fd: make_problem(fd), x0, shape, **skw)
                for k, v in res.items():
                    out[f"{name}_{rname}_{tag}_{k}"] = v
                print(name, rname, tag, "iterations", res["nit"], "f", res["f"], flush=True)
            for rname, (cls, kw) in cruns.items():
                res = solve_all(cls, kw, lambda: make_problem(kind != "sphere"), x0c, shape, ineq_constraints=con)
                for k, v in res.items():
                    out[f"{name}_{rname}_{tag}_{k}"] = v
                print(name, rname, tag, "iterations", res["nit"], "f", res["f"], flush=True)
        torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "tr_traces.npz"), **out)
    print("wrote tr_traces.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB uncompressed")


if __name__ == "__main__":
    main()
