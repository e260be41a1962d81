#!/usr/bin/env python3
"""Per-iteration traces of the reference's ConstrainedTrustRegions with SEVERAL inequality constraints at once: the five bound
constraints of examples/bo_sphere/constrained_benchmark_examples/gabo_sphere_bound_constraints.py:94-121 (x >= 0, |y| <= 0.6, |z| <= 0.6 on
S^2, `ConstrainedTrustRegions(maxiter=200)`, starts drawn inside the box as its `sample_sphere_constrained` does) - the quadratic for the
step to the linearised constraints then runs over the violated subset (constrained_trust_regions.py:569-590), which single-constraint
records never exercise - once with the example's box (the optimum is inside it) and once with a tighter one
("box2": y <= 0.3, z <= 0.05, two bounds active at the solution).  Development container only; needs /root/reference.  Same recording as make_golden_tr_traces.py (imported), the
kernel-mean cost of sph3 in tr_traces.npz, float64.  Also the strict variant.  -> tests/golden/tr_traces_box.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_tr_traces as T  # noqa: E402
base = T.base

BOX = dict(xl=0.0, xu=1.0, yl=-0.6, yu=0.6, zl=-0.6, zu=0.6)


def box_constraints(b=BOX):
    return [lambda x: x[0] - b["xl"], lambda x: x[1] - b["yl"], lambda x: b["yu"] - x[1], lambda x: x[2] - b["zl"], lambda x: b["zu"] - x[2]]


def main():
    g = np.load(os.path.join(HERE, "tr_traces.npz"))
    rng = np.random.default_rng(99)
    out = {}
    torch.set_default_dtype(torch.float64)
    name, n = "sph3", 3
    Yt, wt, beta = torch.tensor(g[f"{name}_Y"]), torch.tensor(g[f"{name}_w"]), float(g[f"{name}_beta"])
    man = base.SphereMan(n)

    def cost(x):
        dd = base.sphere_distance_torch(x[None].double(), Yt)
        return -(wt * torch.exp(-beta * dd * dd)).sum()
    b = BOX
    x0 = []
    while len(x0) < 8:                                   # (:123-131)
        s = np.array([rng.uniform(b["xl"], b["xu"]), rng.uniform(b["yl"], b["yu"]), rng.uniform(b["zl"], b["zu"])])
        s = s / np.linalg.norm(s)
        if s[0] > b["xl"] and b["yl"] < s[1] < b["yu"] and b["zl"] < s[2] < b["zu"]:
            x0.append(s)
    x0 = np.stack(x0)
    out[f"{name}_box_x0"] = x0
    make_problem = lambda: base.Problem(manifold=man, cost=cost, verbosity=0, arg=torch.Tensor())   # noqa: E731
    # the same with a box the unconstrained optimum (0.887, 0.453, 0.093) lies OUTSIDE of in two coordinates: two bounds active at once
    b2 = dict(BOX, yu=0.3, zu=0.05)
    x02 = []
    while len(x02) < 8:
        s = np.array([rng.uniform(b2["xl"], b2["xu"]), rng.uniform(b2["yl"], b2["yu"]), rng.uniform(b2["zl"], b2["zu"])])
        s = s / np.linalg.norm(s)
        if s[0] > b2["xl"] and b2["yl"] < s[1] < b2["yu"] and b2["zl"] < s[2] < b2["zu"]:
            x02.append(s)
    x02 = np.stack(x02)
    out[f"{name}_box2_x0"] = x02
    for rname, cls, starts, bb in (("box", base.ConstrainedTrustRegions, x0, BOX), ("box_strict", base.StrictConstrainedTrustRegions, x0, BOX),
                                   ("box2", base.ConstrainedTrustRegions, x02, b2), ("box2_strict", base.StrictConstrainedTrustRegions, x02, b2)):
        res = T.solve_all(cls, {"maxiter": T.MAXIT}, make_problem, starts, (n,), ineq_constraints=box_constraints(bb))
        for k, v in res.items():
            out[f"{name}_{rname}_f64_{k}"] = v
        print(name, rname, "iterations", res["nit"], "f", res["f"].round(6), "\n  x", res["x"].round(4).tolist(), flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "tr_traces_box.npz"), **out)
    print("wrote tr_traces_box.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB uncompressed")


if __name__ == "__main__":
    main()
