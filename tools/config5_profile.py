"""The measurable pieces of BASELINE config 5 in a loop, for `rocprofv3 --kernel-trace --stats` (kernel durations without the host launch overhead\nthat dominates `tools/config5_bench.py` for the 2-6 us kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
from tools.sweep_bench import mandel
n, D, d = 4096, 20, 2
rng = np.random.default_rng(1234)
q = np.linalg.qr(rng.standard_normal((n, D, D)))[0]
X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.05, 5.0, (n, D)), q); X = 0.5 * (X + X.transpose(0, 2, 1))
W = np.linalg.qr(rng.standard_normal((D, D)))[0][:, :d]
x = torch.tensor(mandel(X), device="cuda"); w = torch.tensor(np.ascontiguousarray(W), device="cuda")
ops.set_error_checking(False)
for _ in range(30):
    y = ops.spd_project(x, w)
    k = ops.spd_ai_pairwise(y, y, beta=0.6 + np.log(2))
    lg = ops.spd_logm_mandel(y)
    k2 = ops.frobenius_pairwise(lg, lg, beta=1.0)
    k3 = ops.nested_spd_gram(x, x, w, 0.6 + np.log(2), 0)          # round 4: projection + factorisation fused, then the Gram launch
    k4 = ops.nested_spd_gram(x, x, w, 1.0, 8)                      # ... projection + logm fused, then the Frobenius Gram
torch.cuda.synchronize()
