"""development: value and gradient of the fused log-Euclidean acquisition evaluation (gabo_spd_acq_eval) for d = 2 ... 8, saved per library
build (GABO_HIP_LIB) - run once per build, then `--compare a b` checks the two files bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        print(k, "bit-identical" if same else f"DIFFERENT max rel {np.abs(a[k] - b[k]).max() / np.abs(a[k]).max():.2e}")
    sys.exit(0)
from gabotorch_amd import models, ops
from gabotorch_amd.fused_acquisition import FusedAcquisition
from gabotorch_amd.kernel_utils.kernels_spd import SpdLogEuclideanGaussianKernel
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec
DEV = "cuda:0"
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)
out = {}
for d in range(2, 9):
    rng = np.random.default_rng(d)
    n = 40
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.2, 3.0, (n, d)), q)
    X = ops.matrix_to_mandel(t(0.5 * (Xm + Xm.transpose(0, 2, 1)))).cpu().numpy()
    y = np.log(np.linalg.eigvalsh(Xm)).sum(1) ** 2 + 0.1 * rng.standard_normal(n)
    kern = SpdLogEuclideanGaussianKernel().double(); kern.lengthscale = torch.tensor(1.4, dtype=torch.float64)
    gp = models.ExactGP(t(X), t(y), kern, outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    fused = FusedAcquisition.build(acq, to_vec, torch.device(DEV))
    q = np.linalg.qr(rng.standard_normal((50, d, d)))[0]
    P = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.5, 2.2, (50, d)), q)
    f, g = fused.cost_egrad(t(0.5 * (P + P.transpose(0, 2, 1))))
    out[f"f{d}"], out[f"g{d}"] = f.cpu().numpy(), g.cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1])
