"""development: is it the output buffer's placement or the device state that slows the sphere Gram after a hipGraph sweep?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops, _lib
n = 4096
srng = np.random.default_rng(1234)
sx = srng.standard_normal((n, 10)); sx /= np.linalg.norm(sx, axis=1, keepdims=True)
st_ = torch.tensor(sx, device="cuda")
out0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
lib = _lib.load()
def measure(tag, out):
    def call():
        lib.gabo_sphere_pairwise(st_.data_ptr(), st_.data_ptr(), out.data_ptr(), 1, n, n, 10, 0, 0, 1.2931, 0, 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(600):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us  (out ptr {out.data_ptr():#x}, stream {torch.cuda.current_stream().cuda_stream:#x})")
measure("fresh, buffer allocated at start", out0)
from tools.sweep_bench import run_sweep
run_sweep("cuda:0", num_restarts=512, hip_graphs=True, batched_rand=True)
torch.cuda.synchronize()
measure("after a hipGraph sweep, SAME buffer", out0)
out1 = torch.empty(n, n, dtype=torch.float64, device="cuda")
measure("after a hipGraph sweep, new buffer", out1)
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    measure("same buffer, another stream", out0)
import gc; gc.collect(); torch.cuda.empty_cache()
measure("after gc + empty_cache, same buffer", out0)
# a trivial graph of our own: does ANY capture do it?
