"""development: the S^9 exact-Hessian sweep of bench.py (acq_sweep_sphere): time and per-restart iteration statistics - python tools/sphere_sweep_stats.py [R ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gabotorch_amd import ops
from tools.sphere_sweep_bench import run
ops.set_error_checking(False)
for R, extra in [(int(a), e) for a in (sys.argv[1:] or ["64", "512"]) for e in ({"device_selection": False}, {}, {"device_rand": True})]:
    kw = dict(approx=False, constrained=False, R=R, raw=4 * R, **extra)
    for _ in range(3):
        run(**kw)
    ts = []
    for _ in range(7):
        dt, val, its, log = run(**kw)
        ts.append(dt)
    it = log["per_restart_iterations"].cpu().numpy()
    srt = np.sort(it)[::-1]
    print(f"R={R} {extra}: median {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f}); EI* {val:.15e}; iterations: max {it.max()}, at maxiter {int((it >= 50).sum())}, "
          f"largest {srt[:8].tolist()}, mean {it.mean():.2f}")
