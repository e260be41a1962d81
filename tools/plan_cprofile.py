"""cProfile of the per-GP set-up of a sweep (FusedAcquisition.build through _native_sweep_plan), development."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import models, ops
from gabotorch_amd.fused_acquisition import FusedAcquisition
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec
from tools.sweep_bench import mandel
ops.set_error_checking(False)
rng = np.random.default_rng(0)
q = np.linalg.qr(rng.standard_normal((50, 5, 5)))[0]
X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(1e-3, 5.0, (50, 5)), q); X = 0.5 * (X + X.transpose(0, 2, 1))
xv = torch.tensor(mandel(X), device="cuda:0"); y = torch.tensor(rng.standard_normal(50), device="cuda:0")
def once():
    kern = SpdAffineInvariantGaussianKernel(beta_min=0.25)
    gp = models.ExactGP(xv, y, kern, outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(-1.0), maximize=False)
    return acq
for _ in range(5):
    FusedAcquisition.build(once(), to_vec, torch.device("cuda:0"))
acqs = [once() for _ in range(50)]
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
for a in acqs:
    FusedAcquisition.build(a, to_vec, torch.device("cuda:0"))
pr.disable()
torch.cuda.synchronize()
print("build: %.1f us each" % ((time.perf_counter() - t0) / 50 * 1e6))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
