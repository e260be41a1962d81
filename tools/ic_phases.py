"""Host-side anatomy of gen_batch_initial_conditions_manifold in the config-4 sweep (development)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gabotorch_amd.manifold_optimization.manifold_optimize as mo
from gabotorch_amd import models
from tools import sweep_bench
T = {}
def wrap(name, fn):
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return inner
mo._draw_raw_samples = wrap("draw", mo._draw_raw_samples)
mo._score_raw_samples = wrap("score", mo._score_raw_samples)
mo._gather_raw_samples = wrap("gather", mo._gather_raw_samples)
mo.initialize_q_batch_nonneg = wrap("select", mo.initialize_q_batch_nonneg)
mo.gen_batch_initial_conditions_manifold = wrap("initial_conditions", mo.gen_batch_initial_conditions_manifold)
mo.FusedAcquisition.build = staticmethod(wrap("fused.build", mo.FusedAcquisition.build))
for _ in range(3):
    sweep_bench.run_sweep("cuda:0", builtin_constraint=True, device_rand=True)
for _ in range(3):
    T.clear()
    dt, *_ = sweep_bench.run_sweep("cuda:0", builtin_constraint=True, device_rand=True)
    print("total %.2f ms" % (dt * 1e3), {k: round(v * 1e3, 3) for k, v in T.items()})
