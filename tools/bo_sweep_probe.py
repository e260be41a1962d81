"""development: the sweeps of a BO loop (examples/gabo_spd.py flow, S^5_++, 512 restarts, surrogate refitted every iteration) one by one - time, host phases,
iteration statistics; `python tools/bo_sweep_probe.py K` also walks sweep K on the Python path with the solve launch's own record and prints what its longest
restarts did (accepted / rejected iterations, radius, tCG stop reasons)."""
import os, sys, time, types, functools
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gabotorch_amd import manifolds, models, ops
from gabotorch_amd._compat import ScaleKernel
from gabotorch_amd.BO_test_functions.test_functions import ackley_function_spd
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
from gabotorch_amd.Riemannian_utils.spd_constraints_utils_torch import max_eigenvalue_constraint_torch
from gabotorch_amd.Riemannian_utils.spd_utils import spd_sample, symmetric_matrix_to_vector_mandel
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch
dev = "cuda:0"
np.random.seed(0); torch.manual_seed(0)
man = manifolds.PositiveDefinite(5); man.min_eig, man.max_eig = 0.001, 5.0
man.rand = types.MethodType(spd_sample, man)
objective = lambda x: ackley_function_spd(x, man)
con = functools.partial(max_eigenvalue_constraint_torch, maximum_eigenvalue=man.max_eig)
x = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(man.rand()) for _ in range(5)]), device=dev)
y = torch.cat([objective(v) for v in x]).reshape(-1).to(dev)
solver = BatchedTrustRegions(mingradnorm=1e-4, maxiter=100)
ops.set_error_checking(False)
STOP = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for it in range(16):
    kern = ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=0.25), outputscale_prior=models.GammaPrior(2.0, 0.15))
    gp = models.SingleTaskGP(x, y, kern, noise_prior=models.GammaPrior(1.1, 0.05))
    models.fit_gpytorch_model(gp)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    if it == STOP:
        # anatomy of this sweep's long restarts: the Python path with the solve launch's own record
        st_np, st_t = np.random.get_state(), torch.get_rng_state()
        solver.trace = []
        joint_optimize_manifold(acq, man, solver, q=1, num_restarts=512, raw_samples=1024, bounds=None,
                                options={"device": dev, "device_rand": True, "native_sweep": False}, inequality_constraints=[con],
                                pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
        tr = solver.trace; solver.trace = None
        its = solver.log["per_restart_iterations"].cpu().numpy()
        X = torch.stack([e["x"] for e in tr]).cpu().numpy()            # iterations x R x d x d
        De = torch.stack([e["Delta"] for e in tr]).cpu().numpy()
        Stp = torch.stack([e["stop_inner"] for e in tr]).cpu().numpy()
        names = ["negcurv", "exceededTR", "lin", "superlin", "maxinner", "modelinc", "constraints"]
        long = list(np.argsort(-its)[:4]) + list(np.argsort(-np.where(its < 100, its, 0))[:10])      # the longest, and the longest that did not hit maxiter
        for r in long:
            k = int(its[r])
            moved = [bool(np.abs(X[j + 1, r] - X[j, r]).max() > 0) for j in range(min(k, len(tr)) - 1)]
            lam = np.linalg.eigvalsh(X[min(k, len(tr)) - 1, r])
            stops = {names[s_]: int((Stp[:k, r] == s_).sum()) for s_ in range(7) if (Stp[:k, r] == s_).any()}
            print(f"  restart {r:3d}: {k:3d} iterations, accepted {sum(moved)}, Delta first {De[0, r]:.3g} last {De[min(k, len(tr)) - 1, r]:.3g}, stops {stops}, lam_max {lam.max():.4f} lam_min {lam.min():.2e}")
        np.random.set_state(st_np); torch.set_rng_state(st_t)
    tl = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nx = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=512, raw_samples=1024, bounds=None,
                                 options={"device": dev, "device_rand": True, "timeline": tl}, inequality_constraints=[con],
                                 pre_processing_manifold=vector_to_symmetric_matrix_mandel_torch,
                                 post_processing_manifold=symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lg = solver.log
    its = lg["per_restart_iterations"].cpu().numpy()
    srt = np.sort(its)[::-1]
    ph = " ".join(f"{n.split()[-1]}={1e3*(t-tl[0][1]):.2f}" for n, t in tl)
    print(f"it {it:2d} n={len(y):2d} sweep {dt*1e3:.2f} ms native={lg.get('native_sweep')} devsel={lg.get('device_selection')} iters max {its.max()} at100 {(its>=100).sum()} top<100 {srt[srt<100][:4].tolist()} mean {its.mean():.1f} | {ph}")
    if os.environ.get("GABO_DUO_TIMES_PROBE"):      # library built with -DGABO_DUO_TIMES (tools/duo_times.py): the longest restarts of this sweep's solve launch
        import ctypes
        from gabotorch_amd import _lib
        buf = (ctypes.c_longlong * (4 * 512))()
        _lib.load().gabo_debug_duo_times(buf, 512)
        a = np.array(buf[:]).reshape(512, 4)
        for k in np.argsort(-a[:, 0])[:4]:
            print(f"      restart {k:3d}: {a[k, 0] / 1e6:6.2f} M cycles, {a[k, 1]:3d} iterations, speculated {a[k, 2]:3d}, not {a[k, 3] % 1000:3d}, applied as scalar updates {a[k, 3] // 1000:3d}")
    ny = objective(nx[0]).reshape(-1).to(dev)
    x, y = torch.cat([x, nx.detach()]), torch.cat([y, ny])
