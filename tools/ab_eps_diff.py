"""Effect of the Gaussian-mode QL threshold alone: the full N = 4096, d = 10 Gram of one build against the 1e-20 build's (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
from tools.dev_bench import spd_set
tag, ref = sys.argv[1], sys.argv[2]
ops.set_error_checking(False)
beta = 0.2 + float(np.log(2.0))
x = torch.tensor(spd_set(4096, 10), device="cuda")
k = ops.spd_ai_pairwise(x, x, beta=beta).cpu().numpy()
if ref == "save":
    np.save("/tmp/k_ref.npy", k)
else:
    r = np.load("/tmp/k_ref.npy")
    with np.errstate(all="ignore"):
        d2, r2 = -np.log(k) / beta, -np.log(r) / beta
    big = r > 1e-200
    print(f"[{tag}] vs 1e-20: max |dK|/K {np.max(np.abs(k - r)[big] / r[big]):.2e}, max |d(d^2)| {np.nanmax(np.abs(d2 - r2)[big]):.2e}, entries that differ {int((k != r).sum())} of {k.size}")
