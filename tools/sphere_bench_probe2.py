"""development: how long does the state left by a burst of fp64 SPD Grams slow the sphere Gram down?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
import bench
n = 4096
srng = np.random.default_rng(1234)
sx = srng.standard_normal((n, 10)); sx /= np.linalg.norm(sx, axis=1, keepdims=True)
st_ = torch.tensor(sx, device="cuda")
x = torch.tensor(bench.synthetic_spd_mandel(4096, 10, 1234), device="cuda")
def chunk(iters):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters):
        ops.sphere_pairwise(st_, st_, beta=1.2931)
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / iters * 1e3
chunk(50)
print("fresh", [round(chunk(100), 1) for _ in range(3)])
for burst in (20, 60, 200):
    for _ in range(burst):
        ops.spd_ai_pairwise(x, x, beta=0.9)
    torch.cuda.synchronize()
    print(f"after {burst} SPD Grams:", [round(chunk(100), 1) for _ in range(12)])
    time.sleep(0.5)
# the other way round: does the SPD Gram suffer after a sphere burst?
def spd(iters):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters):
        ops.spd_ai_pairwise(x, x, beta=0.9)
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / iters
time.sleep(1.0)
print("SPD fresh", [round(spd(10), 3) for _ in range(8)])
