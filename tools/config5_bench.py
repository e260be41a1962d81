"""BASELINE.json config 5 (the measurable part): nested projection S^20_++ -> S^2_++ followed by the kernel Gram, N=4096."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
from tools.dev_bench import timeit
from tools.sweep_bench import mandel
n, D, d = 4096, 20, 2
rng = np.random.default_rng(1234)
q = np.linalg.qr(rng.standard_normal((n, D, D)))[0]
X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.05, 5.0, (n, D)), q); X = 0.5 * (X + X.transpose(0, 2, 1))
W = np.linalg.qr(rng.standard_normal((D, D)))[0][:, :d]
x = torch.tensor(mandel(X), device="cuda"); w = torch.tensor(np.ascontiguousarray(W), device="cuda")
ops.set_error_checking(False)
ms_p = timeit(lambda: ops.spd_project(x, w), 20)
y = ops.spd_project(x, w)
ms_ai = timeit(lambda: ops.spd_ai_pairwise(y, y, beta=0.6 + np.log(2)), 20)
ms_lg = timeit(lambda: ops.spd_logm_mandel(y), 20)
lg = ops.spd_logm_mandel(y)
ms_le = timeit(lambda: ops.frobenius_pairwise(lg, lg, beta=1.0), 20)
print(f"config5 N={n} D={D}->d={d}: project {ms_p*1e3:.1f} us ({n/ms_p*1e3:.3e} matrices/s) | nested AI Gram {ms_ai*1e3:.1f} us ({n*n/ms_ai*1e3:.3e} pairs/s) | "
      f"logm {ms_lg*1e3:.1f} us | log-Euclid Gram {ms_le*1e3:.1f} us ({n*n/ms_le*1e3:.3e} pairs/s)")

# ---- the acquisition sweep of config 5: latent S^2_++, log-Euclidean kernel, StrictConstrainedTrustRegions semantics, eigenvalue box
import time
from gabotorch_amd import manifolds, models
from gabotorch_amd.kernel_utils.kernels_spd import SpdLogEuclideanGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat


def latent_sweep(R=512, raw=2048, n_train=50, graphs=True, fused=True, capture=False, partials=False):
    z = y[:n_train]
    lam = np.linalg.eigvalsh(np.einsum("da,ndc,cb->nab", W, X[:n_train], W))
    f = (np.log(lam / 2.0) ** 2).sum(1)
    kern = SpdLogEuclideanGaussianKernel().double()
    kern.lengthscale = torch.tensor(1.5, dtype=torch.float64)
    gp = models.ExactGP(z, torch.tensor(f, device="cuda"), kern, outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(f.min()), maximize=False)
    man = manifolds.PositiveDefinite(d)
    man.min_eig, man.max_eig = 0.05, 5.0
    man.rand = lambda: (lambda u, l: u @ np.diag(l) @ u.T)(np.linalg.qr(np.random.randn(d, d))[0], 0.05 + 4.95 * np.random.rand(d))
    cons = [lambda m: scut.max_eigenvalue_constraint_torch(m, 5.0), lambda m: scut.min_eigenvalue_constraint_torch(m, 0.05)]
    if partials:        # the way the reference builds them: recognised and evaluated on the device, the solve is one launch
        import functools
        cons = [functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=5.0),
                functools.partial(scut.min_eigenvalue_constraint_torch, minimum_eigenvalue=0.05)]
    np.random.seed(7); torch.manual_seed(7)
    solver = BatchedTrustRegions(mingradnorm=2e-4, maxiter=100, minstepsize=1e-4, strict_constraints=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    best = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=R, raw_samples=raw, bounds=None,
                                   options={"device": "cuda:0", "hip_graphs": graphs, "batched_rand": True, "fused_acquisition": fused, "capture_constraints": capture},
                                   inequality_constraints=cons, pre_processing_manifold=to_mat, post_processing_manifold=to_vec,
                                   approx_hessian=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, float(acq(best[None]).item()), solver.log["iterations"]


for label, kw in (("generic autograd path, eager", dict(graphs=False, fused=False)), ("device-resident TR iteration (single-launch log-Euclidean acquisition), eager", dict(graphs=False)),
                  ("device-resident TR iteration, hipGraphs", dict()),
                  ("device-resident TR iteration, hipGraphs incl. the constraint callables", dict(capture=True)),
                  ("single-launch solve (constraints as functools.partial of the built-ins)", dict(graphs=False, partials=True))):
    latent_sweep(**kw)
    dt, val, its = latent_sweep(**kw)
    print(f"config5 latent sweep (512 restarts, n=50, strict TR, log-Euclid kernel) {label}: {dt*1e3:.1f} ms  EI*={val:.6e}  TR iterations={its}")
