"""development: which earlier phase of bench.py leaves the sphere Gram slower (35.9 us sustained inside bench.py against 30.7 in a fresh process)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
import bench
n = 4096
srng = np.random.default_rng(1234)
sx = srng.standard_normal((n, 10)); sx /= np.linalg.norm(sx, axis=1, keepdims=True)
st_ = torch.tensor(sx, device="cuda")
def measure(tag):
    for _ in range(600):
        ops.sphere_pairwise(st_, st_, beta=1.2931)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ks = ops.sphere_pairwise(st_, st_, beta=1.2931)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us   (allocated {torch.cuda.memory_allocated() >> 20} MB, reserved {torch.cuda.memory_reserved() >> 20} MB, out ptr % 2MB = {ks.data_ptr() % (2 << 20)})")
measure("fresh")
x = bench.synthetic_spd_mandel(4096, 10, 1234)
job = bench.GramJob(x, torch.device("cuda", 0), symmetric=False)
for _ in range(85):
    job.step()
torch.cuda.synchronize()
measure("after the headline Gram job")
sym = bench.GramJob(x, torch.device("cuda", 0), symmetric=True)
for _ in range(25):
    sym.step()
torch.cuda.synchronize()
measure("after the symmetric job")
from tools.sweep_bench import run_sweep
run_sweep("cuda:0", num_restarts=512, batched_rand=True, builtin_constraint=True)
measure("after one single-launch sweep")
run_sweep("cuda:0", num_restarts=512, hip_graphs=True, batched_rand=True)
measure("after a hipGraph sweep")
from tools.sphere_sweep_bench import run as sphere_sweep
sphere_sweep(approx=False, constrained=False, device="cuda:0")
measure("after the sphere sweep")
torch.cuda.empty_cache()
measure("after empty_cache")
