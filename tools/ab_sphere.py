"""A/B of sphere Gram variants (development): GABO_HIP_LIB=... python tools/ab_sphere.py tag [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops, _lib
from tools.dev_bench import timeit
tag = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(0)
s = rng.standard_normal((n, 10)); s /= np.linalg.norm(s, axis=1, keepdims=True)
st = torch.tensor(s, device="cuda")
ms = min(timeit(lambda: ops.sphere_pairwise(st, st, beta=1.29), iters=30, warm=5) for _ in range(4))
print(f"[{tag}] sphere gauss dim=10 N={n}: {ms*1e3:.1f} us  {n*n*8/ms*1e3/1e9:.0f} GB/s written")
