"""development: the fused log-Euclidean / Frobenius evaluation at the largest training sets it accepts, against the separate-launch chain"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _lib, models, ops
from gabotorch_amd.fused_acquisition import FusedAcquisition
from gabotorch_amd.kernel_utils.kernels_spd import SpdFrobeniusGaussianKernel, SpdLogEuclideanGaussianKernel
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec
DEV = "cuda:0"
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)
for d in (2, 5, 8):
    n_max = int(_lib.load().gabo_spd_acq_max_train(d))
    for K in (SpdLogEuclideanGaussianKernel, SpdFrobeniusGaussianKernel):
        for n in (n_max, n_max // 2 + 1, 65, 64, 63, 1):
            rng = np.random.default_rng(n + d)
            q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
            Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.5, 2.0, (n, d)), q)
            X = ops.matrix_to_mandel(t(0.5 * (Xm + Xm.transpose(0, 2, 1))))
            y = t(rng.standard_normal(n))
            kern = K().double(); kern.lengthscale = torch.tensor(1.5, dtype=torch.float64)
            gp = models.ExactGP(X, y, kern, outputscale=1.0, noise=0.5)
            acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
            fused = FusedAcquisition.build(acq, to_vec, torch.device(DEV))
            q = np.linalg.qr(rng.standard_normal((9, d, d)))[0]
            P = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.5, 2.0, (9, d)), q)
            x = t(0.5 * (P + P.transpose(0, 2, 1)))
            assert fused.single_launch, (d, n)
            f1, g1 = fused.cost_egrad(x)
            fused.single_launch = False
            f2, g2 = fused.cost_egrad(x)
            ef = float((f1 - f2).abs().max() / f2.abs().max().clamp(min=1e-300)); eg = float((g1 - g2).abs().max() / g2.abs().max().clamp(min=1e-300))
            print(f"d={d} {K.__name__[3:12]} n={n}: value err {ef:.1e} grad err {eg:.1e}", "" if max(ef, eg) < 1e-8 else "  <-- LOOK")
