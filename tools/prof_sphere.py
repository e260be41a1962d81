import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
s = rng.standard_normal((n, 10)); s /= np.linalg.norm(s, axis=1, keepdims=True)
s = torch.tensor(s, device="cuda")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for _ in range(reps):
    ops.sphere_pairwise(s, s, beta=1.29)
torch.cuda.synchronize()
