"""Config 4 of BASELINE.json: gabo_spd S^5_++ acquisition sweep, 512 restarts (sharded over ranks when launched with torchrun)."""
import functools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import manifolds, models, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat


def mandel(m):
    d = m.shape[-1]
    r, c = [], []
    for k in range(d):
        for i in range(d - k):
            r.append(i); c.append(i + k)
    r, c = np.array(r), np.array(c)
    return np.ascontiguousarray(m[..., r, c] * np.where(r == c, 1.0, 2 ** 0.5))


def run_sweep(device, num_restarts=512, raw_samples=2048, d=5, n_train=50, seed=1234, maxiter=100, hip_graphs=False, batched_rand=False, fused=True, device_tcg=True, device_outer=True, capture_constraints=False, strict=False, device_iteration=True, device_rand=False, device_solve=True, builtin_constraint=False, native_sweep=True, constraint=True, use_rand=False, device_selection=None, log_picked=False):
    rng = np.random.default_rng(seed)
    q = np.linalg.qr(rng.standard_normal((n_train, d, d)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(1e-3, 5.0, (n_train, d)), q)
    X = 0.5 * (X + X.transpose(0, 2, 1))
    man = manifolds.PositiveDefinite(d)
    man.min_eig, man.max_eig = 1e-3, 5.0
    # the objective of config 4: Ackley in the tangent space of 2I (BO_test_functions/test_functions_spd.py:14-69; the
    # package's restatement, pinned by tests/golden/objectives.npz)
    from gabotorch_amd.BO_test_functions.test_functions import ackley_function_spd
    Xv = mandel(X)
    y = np.array([float(ackley_function_spd(torch.tensor(Xv[i:i + 1]), man)) for i in range(n_train)])
    kern = SpdAffineInvariantGaussianKernel(beta_min=0.25)
    gp = models.ExactGP(torch.tensor(Xv, device=device), torch.tensor(y, device=device), kern, outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)

    def rand():
        lam = man.min_eig + (man.max_eig - man.min_eig) * np.random.rand(d)
        u, _ = np.linalg.qr(np.random.randn(d, d))
        return u @ np.diag(lam) @ u.T
    man.rand = rand
    man.rand_batch = lambda k: np.stack([rand() for _ in range(0)]) if k == 0 else manifolds.PositiveDefinite.rand_batch(man, k)
    # host samplers read numpy's global stream: one stream per rank (seed + rank); the device sampler is addressed by the global sample
    # index, so there every rank keeps the SAME seed and the shards add up to exactly the single-rank draw
    import torch.distributed as _dist
    _rank = _dist.get_rank() if (_dist.is_available() and _dist.is_initialized()) else 0
    np.random.seed(seed + (0 if device_rand else _rank)); torch.manual_seed(seed)
    ops.set_error_checking(False)
    device = str(device)
    solver = BatchedTrustRegions(mingradnorm=1e-4, maxiter=maxiter, strict_constraints=strict, use_rand=use_rand)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    best = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=num_restarts, raw_samples=raw_samples, bounds=None,
                                   options={"device": device, "hip_graphs": hip_graphs, "batched_rand": batched_rand, "fused_acquisition": fused, "device_tcg": device_tcg, "device_outer": device_outer, "capture_constraints": capture_constraints, "device_iteration": device_iteration, "device_rand": device_rand, "device_solve": device_solve, "native_sweep": native_sweep, "log_picked": log_picked, **({} if device_selection is None else {"device_selection": device_selection})}, inequality_constraints=None if not constraint else [functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=5.0) if builtin_constraint
                                                           else (lambda x: scut.max_eigenvalue_constraint_torch(x, 5.0))],
                                   pre_processing_manifold=to_mat, post_processing_manifold=to_vec, approx_hessian=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt, best, float(acq(best[None]).item()), solver.log


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dt, best, val, log = run_sweep("cuda:0", num_restarts=R, fused=False)
    print(f"sweep R={R} autograd evaluations: {dt:.3f} s  EI*={val:.6e}")
    dt, best, val, log = run_sweep("cuda:0", num_restarts=R, device_tcg=False)
    print(f"sweep R={R} torch tCG: {dt:.3f} s  EI*={val:.6e} TR iterations={log['iterations']} grad evals={log['grad_evals']}")
    dt, best, val, log = run_sweep("cuda:0", num_restarts=R)
    print(f"sweep R={R}: {dt:.3f} s  {R/dt:.1f} restarts/s  EI*={val:.6e}  TR iterations={log['iterations']} cost evals={log['cost_evals']} grad evals={log['grad_evals']}")
    dt, best, val, log = run_sweep("cuda:0", num_restarts=R)
    print(f"sweep R={R} (warm): {dt:.3f} s  {R/dt:.1f} restarts/s")
    for _ in range(2):
        dt, best, val2, log = run_sweep("cuda:0", num_restarts=R, hip_graphs=True, batched_rand=True)
        print(f"sweep R={R} hipGraph evaluations: {dt:.3f} s  {R/dt:.1f} restarts/s  EI*={val2:.6e} (eager {val:.6e})")
    for _ in range(2):
        dt, best, val2, log = run_sweep("cuda:0", num_restarts=R, hip_graphs=True, batched_rand=True, capture_constraints=True)
        print(f"sweep R={R} hipGraphs incl. constraints: {dt:.3f} s  {R/dt:.1f} restarts/s  EI*={val2:.6e}")
    for _ in range(2):
        dt, best, val2, log = run_sweep("cuda:0", num_restarts=R, batched_rand=True, builtin_constraint=True)
        print(f"sweep R={R} single-launch solve (constraint = functools.partial of the built-in): {dt:.4f} s  {R/dt:.1f} restarts/s  EI*={val2:.6e}")
    for _ in range(2):
        dt, best, val2, log = run_sweep("cuda:0", num_restarts=R, device_rand=True, builtin_constraint=True)
        print(f"sweep R={R} single-launch solve, raw samples drawn on the device: {dt:.4f} s  {R/dt:.1f} restarts/s  EI*={val2:.6e}")
    for _ in range(2):
        dt, best, val2, log = run_sweep("cuda:0", num_restarts=R, hip_graphs=True, device_rand=True, capture_constraints=True)
        print(f"sweep R={R} hipGraphs incl. constraints, raw samples drawn on the device: {dt:.4f} s  {R/dt:.1f} restarts/s  EI*={val2:.6e}")
