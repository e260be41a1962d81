"""development: per-launch durations of the sphere Gram (HIP events around every launch), fresh process"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
n = 4096
srng = np.random.default_rng(1234)
sx = srng.standard_normal((n, 10)); sx /= np.linalg.norm(sx, axis=1, keepdims=True)
st_ = torch.tensor(sx, device="cuda")
def series(k, gap=0.0):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    ev[0].record()
    for i in range(k):
        ops.sphere_pairwise(st_, st_, beta=1.2931)
        ev[i + 1].record()
        if gap:
            torch.cuda.synchronize(); time.sleep(gap)
            ev[i + 1].record() if False else None
    torch.cuda.synchronize()
    return np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(k)])
series(50)
for rep in range(3):
    d = series(600)
    print("600 back-to-back launches: mean %.1f  p10 %.1f p50 %.1f p90 %.1f max %.1f;  per block of 50: %s" % (d.mean(), *np.percentile(d, [10, 50, 90]), d.max(), " ".join("%.0f" % x for x in d.reshape(12, 50).mean(1))))
    time.sleep(0.3)
out = torch.empty(n, n, dtype=torch.float64, device="cuda")
from gabotorch_amd import _lib
lib = _lib.load()
def series_same_buffer(k):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    ev[0].record()
    for i in range(k):
        lib.gabo_sphere_pairwise(st_.data_ptr(), st_.data_ptr(), out.data_ptr(), 1, n, n, 10, 0, 0, 1.2931, 0, 0, torch.cuda.current_stream().cuda_stream)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(k)])
d = series_same_buffer(600)
print("same output buffer, C ABI directly: mean %.1f p50 %.1f p90 %.1f; per block of 50: %s" % (d.mean(), np.percentile(d, 50), np.percentile(d, 90), " ".join("%.0f" % x for x in d.reshape(12, 50).mean(1))))
