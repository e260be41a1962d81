"""Host timeline of the native config-4 sweep (joint_optimize_manifold through gabo_spd_sweep_score / gabo_spd_sweep_solve): wall-clock of
each phase of ONE call, per restart count.  `python tools/sweep_native_phases.py [R ...]`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import sweep_bench
from gabotorch_amd.manifold_optimization import manifold_optimize as mo

orig = mo.joint_optimize_manifold
TL = []


def wrapped(*a, **k):
    k["options"] = dict(k.get("options") or {}, timeline=TL)
    return orig(*a, **k)


sweep_bench.joint_optimize_manifold = wrapped
for R in [int(a) for a in sys.argv[1:]] or [64, 512]:
    kw = dict(num_restarts=R, raw_samples=4 * R, device_rand=True, builtin_constraint=True)
    for _ in range(4):
        sweep_bench.run_sweep("cuda:0", **kw)
    rows = []
    for rep in range(7):
        TL.clear()
        dt = sweep_bench.run_sweep("cuda:0", **kw)[0]
        t0 = TL[0][1]
        rows.append((dt, [(n, t - t0) for n, t in TL]))
    rows.sort(key=lambda r: r[0])
    dt, marks = rows[len(rows) // 2]
    print(f"R={R}: median {dt * 1e3:.3f} ms  (min {rows[0][0] * 1e3:.3f})")
    prev = 0.0
    for n, t in marks:
        print(f"   {t * 1e3:7.3f} ms (+{(t - prev) * 1e3:6.3f})  {n}")
        prev = t
