"""Single-GPU prediction of the strong scaling of the 512-restart acquisition sweep (config 4): times the replicated stage
(initial-condition generation) and rank 0's share of the trust-region stage for world sizes 1, 2, 4, 8."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import manifolds, models, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
from gabotorch_amd.manifold_optimization import manifold_optimize as mo
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat
from tools.sweep_bench import mandel


def setup(device, d=5, n_train=50, seed=1234):
    rng = np.random.default_rng(seed)
    q = np.linalg.qr(rng.standard_normal((n_train, d, d)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(1e-3, 5.0, (n_train, d)), q)
    X = 0.5 * (X + X.transpose(0, 2, 1))
    y = (np.log(np.linalg.eigvalsh(X) / 2.0) ** 2).sum(1)
    kern = SpdAffineInvariantGaussianKernel(beta_min=0.25)
    gp = models.ExactGP(torch.tensor(mandel(X), device=device), torch.tensor(y, device=device), kern, outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    man = manifolds.PositiveDefinite(d)
    man.min_eig, man.max_eig = 1e-3, 5.0
    return acq, man


def main():
    dev = "cuda:0"
    acq, man = setup(dev)
    ops.set_error_checking(False)
    R, raw = 512, 2048
    opts = {"device": dev, "hip_graphs": True, "batched_rand": True}
    res = {}
    for rep in range(3):
        np.random.seed(1234); torch.manual_seed(1234)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ic = mo.gen_batch_initial_conditions_manifold(acq, man, None, None, R, raw, torch.float64, opts, to_vec)
        torch.cuda.synchronize(); res["init_s"] = time.perf_counter() - t0
    for world in (1, 2, 4, 8):
        idx = mo.shard_restarts(R, 0, world)
        best = None
        for rep in range(3):
            solver = BatchedTrustRegions(mingradnorm=1e-4, maxiter=100)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c, v = mo.gen_candidates_manifold(ic[idx], acq, man, solver, to_mat, to_vec, options={"hip_graphs": True},
                                              inequality_constraints=[lambda x: scut.max_eigenvalue_constraint_torch(x, 5.0)],
                                              approx_hessian=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[f"tr_s_world{world}"] = best
        res[f"tr_iters_world{world}"] = solver.log["iterations"]
    for world in (1, 2, 4, 8):
        res[f"predicted_sweep_s_world{world}"] = res["init_s"] + res[f"tr_s_world{world}"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
