"""Acquisition sweep on the sphere (gabo_sphere_bound_constraints-shaped): GP + EI on S^9, 512 restarts, FD Hessian, one bound
constraint x[0] >= 0.1 given as an opaque lambda.  Generic lock-step path vs fused chain (+ hipGraphs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import manifolds, models, ops
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold


def run(dim=10, n_train=50, R=512, raw=2048, graphs=False, fused=True, constrained=True, approx=True, device_tcg=True, maxiter=50, capture=False, device=None, native=True,
        device_selection=None, device_rand=False, log_picked=False):
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n_train, dim)); X /= np.linalg.norm(X, axis=1, keepdims=True)
    y = np.arccos(np.clip(X[:, 0], -1, 1)) ** 2 + 0.05 * rng.standard_normal(n_train)
    gp = models.ExactGP(torch.tensor(X, device=device), torch.tensor(y, device=device), SphereGaussianKernel(beta_min=0.6), outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    man = manifolds.Sphere(dim)
    import torch.distributed as _dist
    _rank = _dist.get_rank() if (_dist.is_available() and _dist.is_initialized()) else 0
    np.random.seed(5 + _rank); torch.manual_seed(5)      # host sampler: one numpy stream per rank (the raw samples are sharded by index)
    cons = [lambda x: x[..., 0] - 0.1] if constrained else None
    solver = BatchedTrustRegions(mingradnorm=1e-5, maxiter=maxiter)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    best = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=R, raw_samples=raw, bounds=None,
                                   options={"device": str(device), "hip_graphs": graphs, "batched_rand": True, "fused_acquisition": fused, "device_tcg": device_tcg, "capture_constraints": capture, "native_sweep": native,
                                            "device_rand": device_rand, "log_picked": log_picked, **({} if device_selection is None else {"device_selection": device_selection})},
                                   inequality_constraints=cons, approx_hessian=approx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, float(acq(best[None]).item()), solver.log["iterations"], solver.log


if __name__ == "__main__":
    ops.set_error_checking(False)
    for label, kw in (("generic autograd, eager", dict(fused=False)), ("fused chain + torch tCG, eager", dict(device_tcg=False)),
                      ("fused chain + torch tCG, hipGraphs", dict(device_tcg=False, graphs=True)),
                      ("device-resident iteration (opaque constraint lambda), eager", dict()),
                      ("device-resident iteration, hipGraphs incl. the constraint", dict(graphs=True, capture=True)),
                      ("unconstrained, FD Hessian: single-launch solve", dict(constrained=False)),
                      ("unconstrained, FD Hessian: torch tCG + hipGraphs", dict(constrained=False, device_tcg=False, graphs=True)),
                      ("exact Hessian (autograd double backward), unconstrained, generic", dict(approx=False, constrained=False, fused=False)),
                      ("exact Hessian (closed form on the device), unconstrained: single-launch solve", dict(approx=False, constrained=False)),
                      ("exact Hessian (closed form on the device), bound constraint lambda", dict(approx=False))):
        run(**kw)
        dt, val, its, _ = run(**kw)
        print(f"sphere sweep S^9 512 restarts {label}: {dt*1e3:.1f} ms  EI*={val:.6e}  TR iterations={its}")
