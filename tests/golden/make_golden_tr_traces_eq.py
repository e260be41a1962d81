#!/usr/bin/env python3
"""Per-iteration traces of the reference's ConstrainedTrustRegions / StrictConstrainedTrustRegions with EQUALITY constraints on the
sphere (development container only; needs /root/reference) - the setting of examples/bo_sphere/constrained_benchmark_examples/
gabo_sphere_equality_constraints.py:100-118 (the great circle x[1] = yc, starts drawn ON the constraint as its
`sample_sphere_constrained` does), which tr_traces.npz does not hold.  Same recording and packing as make_golden_tr_traces.py (imported),
same kernel-mean costs (Y, w, beta read from tr_traces.npz), float64 default dtype only.

Runs per sphere (S^2 in R^3, S^4 in R^5):
  * "eq":     equality x[1] - 0 = 0, starts on the circle                          (ConstrainedTrustRegions, exact Hessian)
  * "eq_fd":  the same with get_hessianfd
  * "eqoff":  equality x[1] - 0.2 = 0, starts OFF that circle (|violation| ~ 0.1..0.4): the tCG exit "reached constraints" with an
              equality term that is not zero at x
  * (equality and inequality constraints TOGETHER are not recorded: the reference indexes the violated inequalities as
    `np.where(const_term < 0)[0] + neq_cons` over the FULL constraint vector (constrained_trust_regions.py:575-579), which picks the wrong
    rows or raises IndexError - it did, on S^4 - so there is no reference behaviour to pin; the package implements the comment above
    that line: equalities + the violated inequalities)
  * "eq_strict": StrictConstrainedTrustRegions with the equality, starts on the circle
-> tests/golden/tr_traces_eq.npz
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


def on_circle(rng, k, n, yc):
    """gabo_sphere_equality_constraints.py:108-115 (`sample_sphere_constrained`), with a seeded generator"""
    x = rng.standard_normal((k, n))
    idx = np.arange(n) != 1
    x[:, 1] = yc
    x[:, idx] = x[:, idx] / np.linalg.norm(x[:, idx], axis=1, keepdims=True) * np.sqrt(1 - yc ** 2)
    return x


def main():
    g = np.load(os.path.join(HERE, "tr_traces.npz"))
    rng = np.random.default_rng(2025)
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

        def make_problem(fd, cost=cost, man=man):
            p = base.Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())
            if fd:
                p._hess = types.MethodType(base.get_hessianfd, p)
            return p
        x_on = on_circle(rng, 4, n, 0.0)
        x_off = rng.standard_normal((4, n)); x_off /= np.linalg.norm(x_off, axis=1, keepdims=True)
        out[f"{name}_eq_x0"], out[f"{name}_eqoff_x0"] = x_on, x_off
        eq0 = [lambda x: x[1] - 0.0]
        eq2 = [lambda x: x[1] - 0.2]
        kw = {"mingradnorm": 1e-6, "maxiter": T.MAXIT}
        runs = {"eq": (base.ConstrainedTrustRegions, False, x_on, dict(eq_constraints=eq0)),
                "eq_fd": (base.ConstrainedTrustRegions, True, x_on, dict(eq_constraints=eq0)),
                "eqoff": (base.ConstrainedTrustRegions, False, x_off, dict(eq_constraints=eq2)),
                "eq_strict": (base.StrictConstrainedTrustRegions, False, x_on, dict(eq_constraints=eq0))}
        for rname, (cls, fd, x0, skw) in runs.items():
            res = T.solve_all(cls, kw, lambda fd=fd: make_problem(fd), x0, (n,), **skw)
            for k, v in res.items():
                out[f"{name}_{rname}_f64_{k}"] = v
            print(name, rname, "iterations", res["nit"], "f", res["f"], "x[1]", res["x"][:, 1], flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "tr_traces_eq.npz"), **out)
    print("wrote tr_traces_eq.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB uncompressed")


if __name__ == "__main__":
    main()
