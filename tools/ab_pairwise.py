"""A/B of pairwise-kernel variants (development): time + worst error against the vectorised oracle on a sub-block.
Usage: GABO_HIP_LIB=... python tools/ab_pairwise.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops, _lib
from tools.dev_bench import spd_set, timeit
from oracle import spd as ospd, sphere as osph

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(_lib.LIB_PATH)
ops.set_error_checking(False)
n, beta = 4096, 0.2 + float(np.log(2.0))
for d in tuple(int(v) for v in os.environ.get("GABO_AB_DIMS", "5,10,12").split(",")):
    xs = spd_set(n, d)
    x = torch.tensor(xs, device="cuda")
    ms = min(timeit(lambda: ops.spd_ai_pairwise(x, x, beta=beta), iters=10, warm=3) for _ in range(3))
    dist = ops.spd_ai_pairwise(x[:192], x[3000:3384], mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    m1, m2 = ospd.vector_to_symmetric_matrix_mandel(xs[:192]), ospd.vector_to_symmetric_matrix_mandel(xs[3000:3384])
    want = ospd.affine_invariant_distance(m1, m2)
    err = float(np.max(np.abs(dist ** 2 - want ** 2) / want ** 2))
    print(f"[{tag}] SPD d={d} N={n}: {ms:.3f} ms  {n*n/ms*1e3:.3e} pairs/s   max rel err of d^2 vs oracle {err:.2e}")
if os.environ.get("GABO_AB_DIMS"):
    sys.exit(0)
rng = np.random.default_rng(0)
s = rng.standard_normal((n, 10)); s /= np.linalg.norm(s, axis=1, keepdims=True)
st = torch.tensor(s, device="cuda")
for mode, name in ((_lib.GABO_OUT_GAUSSIAN, "gauss"), (_lib.GABO_OUT_DISTANCE, "dist"), (_lib.GABO_OUT_LAPLACE, "laplace")):
    ms = min(timeit(lambda: ops.sphere_pairwise(st, st, beta=1.29, mode=mode), iters=20, warm=3) for _ in range(3))
    print(f"[{tag}] sphere {name} dim=10 N={n}: {ms*1e3:.1f} us  {n*n/ms*1e3:.3e} pairs/s  {n*n*8/ms*1e3/1e9:.0f} GB/s written")
k = ops.sphere_pairwise(st[:256], st[2000:2700], beta=1.29).cpu().numpy()
w = osph.sphere_gaussian_kernel(s[:256], s[2000:2700], 1.29)
print(f"[{tag}] sphere gauss max rel err vs oracle {np.max(np.abs(k - w) / w):.2e}")
