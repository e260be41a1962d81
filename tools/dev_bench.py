"""Development timing of the pairwise kernels through the C ABI (not the graded bench; see bench.py)."""
import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import ops

def spd_set(n, d, seed=1234):
    rng = np.random.default_rng(seed)
    out = np.empty((n, d, d))
    for k in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        m = (q * rng.uniform(0.05, 5.0, d)) @ q.T
        out[k] = 0.5 * (m + m.T)
    from tools.sweep_bench import mandel
    return mandel(out)

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ops.set_error_checking(False)
    x = torch.tensor(spd_set(n, d), device="cuda")
    beta = 0.2 + 0.6931472
    ms = timeit(lambda: ops.spd_ai_pairwise(x, x, beta=beta))
    print(f"SPD d={d} N={n}: {ms:.3f} ms  {n*n/ms*1e3:.3e} pairs/s")
    ms = timeit(lambda: ops.spd_ai_pairwise(x, x, beta=beta, symmetric=True))
    print(f"SPD d={d} N={n} symmetric: {ms:.3f} ms  {n*n/ms*1e3:.3e} pairs/s")
    # (parity is checked by tests/ and by bench.py's gate; this tool only times)
    rng = np.random.default_rng(0)
    s = rng.standard_normal((n, 10)); s /= np.linalg.norm(s, axis=1, keepdims=True)
    s = torch.tensor(s, device="cuda")
    ms = timeit(lambda: ops.sphere_pairwise(s, s, beta=1.29))
    print(f"sphere dim=10 N={n}: {ms:.3f} ms  {n*n/ms*1e3:.3e} pairs/s  {n*n*8/ms*1e3/1e9:.1f} GB/s written")
    ms = timeit(lambda: ops.sphere_pairwise(s, s, beta=1.29, symmetric=True))
    print(f"sphere dim=10 N={n} symmetric: {ms:.3f} ms  {n*n/ms*1e3:.3e} pairs/s")
