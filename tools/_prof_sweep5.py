import os, sys, cProfile, pstats, time, types, functools
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import hd_gabo_spd as H
from hd_gabo_spd import *
dim, latent, device = 20, 2, "cuda:0"
np.random.seed(1); torch.manual_seed(1)
big = manifolds.PositiveDefinite(dim); big.min_eig, big.max_eig = 0.1, 5.0
big.rand = types.MethodType(spd_sample, big)
small = manifolds.PositiveDefinite(latent); small.min_eig, small.max_eig = 0.1, 5.0
x_data = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(big.rand()) for _ in range(10)]), device=device)
y = torch.cat([rosenbrock_function_spd(x, big) for x in x_data]).reshape(-1).to(device)
y_std = (y - y.mean()) / y.std()
R = np.linalg.qr(np.random.randn(dim, dim))[0]
W = torch.tensor(R[:, :latent], device=device); V = torch.tensor(R[:, latent:], device=device)
qc = np.linalg.qr(np.random.randn(dim-latent, dim-latent))[0]
bottom = torch.tensor((qc * np.random.uniform(0.5, 2.0, dim-latent)) @ qc.T, device=device)
K0 = np.random.randn(latent, dim-latent); contraction = torch.tensor(0.4 * K0 / np.linalg.norm(K0), device=device)
z_data = ops.spd_project(x_data, W)
solver = BatchedTrustRegions(mingradnorm=2e-4, maxiter=100, minstepsize=1e-4, strict_constraints=True)
ops.set_error_checking(False)
def go(restarts=5, raw=100):
    latent_kernel = SpdLogEuclideanGaussianKernel().double()
    small.rand = types.MethodType(functools.partial(random_nested_spd_with_spd_eigenvalue_constraints, random_spd_fct=big.rand, projection_matrix=W), small)
    cons = [functools.partial(max_eigenvalue_nested_spd_constraint, maximum_eigenvalue=big.max_eig, projection_matrix=W, projection_complement_matrix=V, bottom_spd_matrix=bottom, contraction_matrix=contraction),
            functools.partial(min_eigenvalue_nested_spd_constraint, minimum_eigenvalue=big.min_eig, projection_matrix=W, projection_complement_matrix=V, bottom_spd_matrix=bottom, contraction_matrix=contraction)]
    gp = models.ExactGP(z_data, y_std, latent_kernel, outputscale=1.0, noise=0.01, mean=0.0)
    acq = models.ExpectedImprovement(gp, best_f=float(y_std.min()), maximize=False)
    z = joint_optimize_manifold(acq, small, solver, q=1, num_restarts=restarts, raw_samples=raw, bounds=None, options={"device": device, "hip_graphs": True}, inequality_constraints=cons,
                                pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch, post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
    torch.cuda.synchronize(); return z
go(); go()
for r, raw in ((5, 100), (512, 1024)):
    go(r, raw)
    t = time.perf_counter(); go(r, raw); print("restarts", r, "raw", raw, "wall", time.perf_counter() - t, "iterations", solver.log["iterations"])
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
