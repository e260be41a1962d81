"""development A/B: the d = 2 Gaussian Gram (spd_ai_gauss2_kernel) at N = 4096, sustained and cold - python tools/ab_gauss2.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
from tools.dev_bench import spd_set
tag = sys.argv[1] if len(sys.argv) > 1 else "main"
ops.set_error_checking(False)
n = 4096
x = torch.tensor(spd_set(n, 2), device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ref = ops.spd_ai_pairwise(x, x, beta=0.7)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for c in range(20):
        ops.spd_ai_pairwise(x, x, beta=0.7)
        ev[c + 1].record()
    torch.cuda.synchronize()
    cold = [ev[c].elapsed_time(ev[c + 1]) for c in range(20)]
    for _ in range(600):
        ops.spd_ai_pairwise(x, x, beta=0.7)
    blocks = []
    for b in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ops.spd_ai_pairwise(x, x, beta=0.7)
        e1.record()
        torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / 100)
print(f"[{tag}] d=2 Gaussian Gram N={n} (prep + gauss2 launches): sustained median {np.median(blocks) * 1e3:.2f} us (blocks {[round(b * 1e3, 2) for b in blocks]}), cold median {np.median(cold) * 1e3:.2f} us, checksum {float(ref.sum()):.12e}")
