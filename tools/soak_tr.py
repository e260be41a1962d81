"""Soak of the device trust regions against the generic lock-step solver (the one tests pin to the reference's records iterate by iterate):
random surrogates (S^d_++ affine-invariant / log-Euclidean Gaussian kernel, sphere Gaussian kernel; EI or posterior mean), random eigenvalue
bounds (none / max / box; plain or strict), random starts - every outer iteration of every restart of the single-launch solve (its own
record, gabo_tr_solve_record) and of the propose / update launches walked against the generic path's trace: radius (exact), tCG stop reason,
iterate within `atol`.  Prints the share of restart-iterations that agree and where runs part.

    python tools/soak_tr.py [--cases 60] [--seed 0]
"""
import argparse
import functools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import manifolds, models, ops                                                       # noqa: E402
from gabotorch_amd.kernel_utils.kernels_spd import (SpdAffineInvariantGaussianKernel, SpdFrobeniusGaussianKernel,  # noqa: E402
                                                    SpdLogEuclideanGaussianKernel)
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel                             # noqa: E402
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions              # noqa: E402
from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold              # noqa: E402
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut                         # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch as to_vec,  # noqa: E402
                                                            vector_to_symmetric_matrix_mandel_torch as to_mat)

DEV = "cuda:0"
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)      # noqa: E731


def walk(ref, got, atol):
    """-> (restart-iterations compared, in agreement, list of (restart, iteration, what) where runs part)"""
    R = ref[0]["x"].shape[0]
    total = agree = 0
    parted = []
    for s in range(R):
        for k in range(min(len(ref), len(got))):
            a, b = ref[k], got[k]
            if not bool(a["active"][s]):
                break
            total += 1
            if not bool(b["active"][s]):
                parted.append((s, k, "ended early"))
                break
            dx = float((a["x"][s] - b["x"][s]).abs().max())
            if float(a["Delta"][s]) != float(b["Delta"][s]) or int(a["stop_inner"][s]) != int(b["stop_inner"][s]) or dx > atol:
                parted.append((s, k, f"Delta {float(a['Delta'][s]):.3g}/{float(b['Delta'][s]):.3g} stop {int(a['stop_inner'][s])}/{int(b['stop_inner'][s])} dx {dx:.1e}"))
                break
            agree += 1
    return total, agree, parted


def spd_case(rng):
    d = int(rng.choice([2, 3, 4, 5, 6, 7, 8]))
    n = int(rng.integers(4, 60))
    flav = int(rng.integers(0, 5))          # 0, 1: affine-invariant; 2, 3: log-Euclidean; 4: Frobenius
    le, frob = flav in (2, 3), flav == 4
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.2, 3.0, (n, d)), q)
    X = ops.matrix_to_mandel(t(0.5 * (Xm + Xm.transpose(0, 2, 1)))).cpu().numpy()
    y = np.log(np.linalg.eigvalsh(Xm)).sum(1) ** 2 + 0.1 * rng.standard_normal(n)
    if le or frob:
        kern = (SpdLogEuclideanGaussianKernel if le else SpdFrobeniusGaussianKernel)().double()
        kern.lengthscale = torch.tensor(float(rng.uniform(0.8, 2.0)), dtype=torch.float64)
    else:
        kern = SpdAffineInvariantGaussianKernel(beta_min=float(rng.uniform(0.2, 0.8)))
    gp = models.ExactGP(t(X), t(y), kern, outputscale=float(rng.uniform(0.5, 2.0)), noise=1e-2)
    acq = (models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False) if rng.integers(0, 3) else models.PosteriorMean(gp, maximize=False))
    R = 24
    q = np.linalg.qr(rng.standard_normal((R, d, d)))[0]
    P = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.5, 2.2, (R, d)), q)
    x0 = ops.matrix_to_mandel(t(0.5 * (P + P.transpose(0, 2, 1))))[:, None]
    kind = int(rng.integers(0, 3))
    strict = bool(rng.integers(0, 2)) and kind > 0
    cons = None
    if kind >= 1:
        cons = [functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=float(rng.uniform(2.3, 3.0)))]
    if kind == 2:
        cons.append(functools.partial(scut.min_eigenvalue_constraint_torch, minimum_eigenvalue=float(rng.uniform(0.2, 0.45))))
    desc = f"S^{d}_++ {'LE' if le else 'Frob' if frob else 'AI'} n={n} {type(acq).__name__} cons={kind} strict={strict}"
    return desc, acq, manifolds.PositiveDefinite(d), x0, cons, strict, dict(pre_processing_manifold=to_mat, post_processing_manifold=to_vec, approx_hessian=True)


def sphere_case(rng):
    dim = int(rng.choice([3, 4, 6, 10, 21, 51, 101]))            # (the reference's gabo_sphere.py runs up to dim = 100)
    n = int(rng.integers(4, 300 if rng.integers(0, 4) == 0 else 60))
    X = rng.standard_normal((n, dim)); X /= np.linalg.norm(X, axis=1, keepdims=True)
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
    gp = models.ExactGP(t(X), t(y), SphereGaussianKernel(beta_min=float(rng.uniform(0.5, 6.5))), outputscale=float(rng.uniform(0.5, 2.0)), noise=1e-2)
    acq = (models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False) if rng.integers(0, 3) else models.PosteriorMean(gp, maximize=False))
    R = 24
    P = rng.standard_normal((R, dim)); P /= np.linalg.norm(P, axis=1, keepdims=True)
    fd = bool(rng.integers(0, 2))
    desc = f"S^{dim - 1} n={n} {type(acq).__name__} {'FD' if fd else 'exact'} Hessian"
    return desc, acq, manifolds.Sphere(dim), t(P)[:, None], None, False, dict(approx_hessian=fd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--atol", type=float, default=1e-6)
    ap.add_argument("--only", type=int, default=-1, help="run this case alone (the generator is advanced through the earlier ones)")
    ap.add_argument("--no-record", action="store_true", help="(with --only) the single-launch solve without its record")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    tot = {"solve": [0, 0], "plan": [0, 0]}
    worst = []
    ops.set_error_checking(False)
    for c in range(a.cases):
        desc, acq, man, x0, cons, strict, kw = (spd_case if c % 3 else sphere_case)(rng)
        if a.only >= 0 and c != a.only:
            continue
        traces = {}
        if os.environ.get("GABO_SOAK_VERBOSE"):
            print("case", c, desc, flush=True)
        for name, opts in (("torch", {"device_tcg": False}), ("solve", {}), ("plan", {"device_solve": False})):
            solver = BatchedTrustRegions(mingradnorm=1e-5, maxiter=30, strict_constraints=strict)
            solver.trace = None if (a.no_record and name == "solve") else []
            if os.environ.get("GABO_SOAK_VERBOSE"):
                print("   plan", name, flush=True)
            gen_candidates_manifold(x0, acq, man, solver, inequality_constraints=cons, options=opts, **kw)
            traces[name] = solver.trace if solver.trace is not None else traces["torch"]
            if name == "solve" and "one_launch_solve" not in solver.log:
                desc += " [no single-launch form]"
        line = [f"{c:3d} {desc:58s}"]
        for name in ("solve", "plan"):
            total, agree, parted = walk(traces["torch"], traces[name], a.atol)
            tot[name][0] += total
            tot[name][1] += agree
            line.append(f"{name}: {agree}/{total}")
            if parted:
                worst.append((c, desc, name, parted[:3]))
        print("  ".join(line), flush=True)
    for name, (total, agree) in tot.items():
        print(f"{name}: {agree} of {total} restart-iterations agree with the generic path ({100.0 * agree / max(total, 1):.2f} %)")
    for w in worst[:40]:
        print("parted:", w)


if __name__ == "__main__":
    main()
