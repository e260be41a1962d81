"""Runs the SPD Gram build a few times so rocprofv3 (--kernel-trace / --pmc) has something to look at."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gabotorch_amd import ops
from tools.dev_bench import spd_set
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sym = len(sys.argv) > 3 and sys.argv[3] == "sym"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ops.set_error_checking(False)
x = torch.tensor(spd_set(n, d), device="cuda")
for _ in range(reps):
    ops.spd_ai_pairwise(x, x, beta=0.2 + 0.6931472, symmetric=sym)
torch.cuda.synchronize()
