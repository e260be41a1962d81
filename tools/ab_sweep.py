"""development A/B of the config-4 sweep (native driver): time and per-restart iteration statistics - python tools/ab_sweep.py [tag] [R ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.sweep_bench import run_sweep
tag = sys.argv[1] if len(sys.argv) > 1 else "main"
for R in [int(a) for a in sys.argv[2:]] or [64, 512]:
    kw = dict(num_restarts=R, raw_samples=4 * R, device_rand=True, builtin_constraint=True)
    for _ in range(4):
        run_sweep("cuda:0", **kw)
    ts, val, log = [], None, None
    for _ in range(9):
        dt, _, val, log = run_sweep("cuda:0", **kw)
        ts.append(dt)
    it = log["per_restart_iterations"].numpy()
    srt = np.sort(it)[::-1]
    print(f"[{tag}] R={R}: median {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f}); EI* {val:.15e}; iterations: max {it.max()}, at maxiter {int((it >= 100).sum())}, "
          f"largest below maxiter {srt[srt < 100][:5].tolist()}, mean {it.mean():.2f}; sum of final costs {float(log['final_cost'].sum()):.17e}, sum of iterations {int(it.sum())}")
    try:      # the two-wave solve's speculation counters (spd_tr_solve_duo.hip), when that kernel ran
        import ctypes
        from gabotorch_amd import _lib
        h, m = ctypes.c_longlong(0), ctypes.c_longlong(0)
        _lib.load().gabo_spd_tr_two_waves_counters(ctypes.byref(h), ctypes.byref(m), 1)
        if h.value + m.value:
            print(f"[{tag}] R={R}: two-wave solve: {h.value} iterations with the speculated step, {m.value} without ({h.value / (h.value + m.value):.3f})")
    except Exception as e:
        print("no counters:", e)
