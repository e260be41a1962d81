"""BASELINE.json config 5 (the measurable part): nested projection S^20_++ -> S^2_++ followed by the kernel Gram, N=4096."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import _lib, ops
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
