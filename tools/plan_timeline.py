import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import sweep_bench
from gabotorch_amd import ops, models, fused_acquisition as fa
from gabotorch_amd.manifold_optimization import manifold_optimize as mo
TL = []
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def w(*a, **k):
        TL.append(("-> " + (label or name), time.perf_counter()))
        r = f(*a, **k)
        TL.append(("<- " + (label or name), time.perf_counter()))
        return r
    setattr(obj, name, w)
wrap(mo, "_native_sweep_plan")
wrap(fa.FusedAcquisition, "_build", "Fused._build")
wrap(fa, "_surrogate_view")
wrap(ops, "spd_gp_prepare")
lib = ops._lib.load()
class L:
    def __getattr__(self, n):
        f = getattr(lib, n)
        def w(*a):
            TL.append(("-> C " + n, time.perf_counter())); r = f(*a); TL.append(("<- C " + n, time.perf_counter())); return r
        return w
ops._lib.load = lambda: L()
mo_lib = L()
kw = dict(num_restarts=64, raw_samples=256, device_rand=True, builtin_constraint=True)
for _ in range(5):
    sweep_bench.run_sweep("cuda:0", **kw)
for rep in range(3):
    TL.clear()
    dt = sweep_bench.run_sweep("cuda:0", **kw)[0]
    # only events after the last "-> _native_sweep_plan"
    i0 = max(i for i, (n, t) in enumerate(TL) if n == "-> _native_sweep_plan")
    t0 = TL[i0][1]
    print(f"--- sweep {dt*1e3:.3f} ms")
    prev = t0
    for n, t in TL[i0:]:
        print(f"  {1e6*(t-t0):8.1f} us (+{1e6*(t-prev):6.1f})  {n}")
        prev = t
