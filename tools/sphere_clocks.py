"""Per-wave timeline of the sphere Gram kernel (development): build with
    python tools/ab_build.py clk sphere_pairwise.hip -DGABO_SPH_CLOCKS [-DGABO_SPH_PROBE=2: without the result stores]
and run  GABO_HIP_LIB=gabotorch_amd/libgabo_hip_clk.so python tools/sphere_clocks.py [N].
Each wave records s_memrealtime (100 MHz) at entry, at the start of its chunk loop and at exit behind the result matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
s = rng.standard_normal((n, 10)); s /= np.linalg.norm(s, axis=1, keepdims=True)
st = torch.tensor(s, device="cuda")
from gabotorch_amd import _lib
lib = _lib.load()
waves = (n // 256) * (n // 64) * 4
buf = torch.zeros(n * n + waves * 4, dtype=torch.float64, device="cuda")
for _ in range(20):
    rc = lib.gabo_sphere_pairwise(st.data_ptr(), st.data_ptr(), buf.data_ptr(), 1, n, n, 10, 0, 0, 1.29, _lib.GABO_OUT_GAUSSIAN, 0,
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
rec = buf[n * n:].reshape(waves, 4).cpu().numpy()
t0 = rec[:, 0].min()
start, loop, end = [(rec[:, k] - t0) / 100.0 for k in range(3)]
q = lambda v: " ".join(f"{x:7.2f}" for x in np.percentile(v, [0, 10, 50, 90, 100]))
print(f"N={n}: {waves} waves; us relative to the first wave's entry; percentiles 0 10 50 90 100")
print("entry            ", q(start))
print("prologue (entry -> chunk loop)", q(loop - start))
print("chunk loop       ", q(end - loop))
print("exit             ", q(end))
print(f"kernel span (first entry -> last exit) {end.max():.2f} us; shader clock while a wave lives: median {np.median(rec[:, 3] / (rec[:, 2] - rec[:, 0]) * 0.1):.3f} GHz")
