"""Gram throughput of the SPD affine-invariant kernel versus the matrix dimension (N = 4096, all N^2 pairs; forward and backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gabotorch_amd import ops
from tools.dev_bench import spd_set, timeit
ops.set_error_checking(False)
n = 4096
for d in (2, 3, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20):
    nn = n if d <= 16 else 512          # the wave-per-pair fallback (forward d > 16, backward d > 12) is a correctness path
    x = torch.tensor(spd_set(nn, d), device="cuda")
    ms = timeit(lambda: ops.spd_ai_pairwise(x, x, beta=0.5), 5 if d > 12 else 10)
    nb = nn if d <= 16 else 512      # (register-resident backward up to d = 16: two lanes per pair from d = 12)
    xb = x[:nb]
    go = torch.ones(nb, nb, dtype=torch.float64, device="cuda")
    msb = timeit(lambda: ops.spd_ai_backward(xb, xb, go, 0.5), 3)
    print(f"d={d:2d} N={nn}: forward {ms:8.3f} ms {nn*nn/ms*1e3:.3e} pairs/s | backward (N={nb}) {msb:8.3f} ms {nb*nb/msb*1e3:.3e} pairs/s")
