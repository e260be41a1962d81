"""development: the sphere Gram timed the way bench.py does it, alone and after an SPD Gram burst"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
import bench
n = 4096
srng = np.random.default_rng(1234)
sx = srng.standard_normal((n, 10)); sx /= np.linalg.norm(sx, axis=1, keepdims=True)
st_ = torch.tensor(sx, device="cuda")
sbeta = 0.6 + float(np.log(2.0))
def run(tag, iters=20):
    for _ in range(5):
        ops.sphere_pairwise(st_, st_, beta=sbeta)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters):
        ks = ops.sphere_pairwise(st_, st_, beta=sbeta)
    s1.record(); torch.cuda.synchronize()
    print(tag, s0.elapsed_time(s1) / iters * 1e3, "us")
run("fresh process")
run("again")
x = torch.tensor(bench.synthetic_spd_mandel(4096, 10, 1234), device="cuda")
for _ in range(60):
    ops.spd_ai_pairwise(x, x, beta=0.9)
torch.cuda.synchronize()
run("after 60 SPD Grams (140 ms of fp64 load)")
run("again")
import time; time.sleep(1.0)
run("after 1 s idle")
big = [torch.empty(1 << 28, dtype=torch.uint8, device="cuda") for _ in range(8)]
run("with 2 GB more allocated")
del big
time.sleep(1.0)
for it in (20, 200, 2000, 2000, 20):
    run(f"back to back x{it}", it)
