"""Backward-kernel timing + error against the oracle (development): python tools/ab_backward.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops, _lib
from tools.dev_bench import spd_set, timeit
from oracle import spd as ospd
tag = sys.argv[1] if len(sys.argv) > 1 else "main"
DIMS = tuple(int(v) for v in os.environ["GABO_AB_DIMS"].split(",")) if "GABO_AB_DIMS" in os.environ else ((10,) if tag == "prof" else (8, 9, 10, 11, 12))
ops.set_error_checking(False)
n, beta = 4096, 0.2 + float(np.log(2.0))
for d in DIMS:
    xs = spd_set(n, d)
    x = torch.tensor(xs, device="cuda")
    go = torch.ones(n, n, dtype=torch.float64, device="cuda")
    fn = lambda: ops.spd_ai_backward(x, x, go, beta=beta) if hasattr(ops, "spd_ai_backward") else None
    xr = x.clone().requires_grad_(True)
    def run():
        k = ops.spd_ai_kernel(xr, x, beta)
        (g,) = torch.autograd.grad(k.sum(), xr)
        return g
    ms = min(timeit(run, iters=3, warm=1) for _ in range(2))
    fw = min(timeit(lambda: ops.spd_ai_pairwise(x, x, beta=beta), iters=5, warm=1) for _ in range(2))
    # error on a small problem against the oracle's closed form
    m = 96
    rng = np.random.default_rng(d)
    gk = rng.standard_normal((m, m))
    x1 = x[:m].clone().requires_grad_(True)
    k = ops.spd_ai_kernel(x1, x[m:2 * m], beta)
    (g1,) = torch.autograd.grad((k * torch.tensor(gk, device="cuda")).sum(), x1)
    want = ospd.spd_ai_gaussian_kernel_grads(xs[:m], xs[m:2 * m], beta, gk)[0]
    err = float(np.max(np.abs(g1.cpu().numpy() - want)) / np.max(np.abs(want)))
    print(f"[{tag}] d={d}: forward+backward {ms:.2f} ms, forward {fw:.2f} ms -> backward ~{ms - fw:.2f} ms ({n*n/(ms-fw)*1e3:.3e} pairs/s)   max err / max|grad| {err:.2e}")
