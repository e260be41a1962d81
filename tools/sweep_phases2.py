"""Host-side anatomy of the single-launch config-4 sweep (development): initial conditions / candidate generation / the rest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gabotorch_amd.manifold_optimization.manifold_optimize as mo
from gabotorch_amd.manifold_optimization import batched_trust_regions as btr
from tools import sweep_bench
T = {}
def wrap(name, fn):
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return inner
mo.gen_batch_initial_conditions_manifold = wrap("initial_conditions", mo.gen_batch_initial_conditions_manifold)
mo.gen_candidates_manifold = wrap("gen_candidates", mo.gen_candidates_manifold)
btr.BatchedTrustRegions.solve = wrap("  solver.solve", btr.BatchedTrustRegions.solve)
for kw in (dict(device_rand=True), dict(batched_rand=True)):
    for _ in range(3):
        sweep_bench.run_sweep("cuda:0", builtin_constraint=True, **kw)
    for _ in range(3):
        T.clear()
        dt, *_ = sweep_bench.run_sweep("cuda:0", builtin_constraint=True, **kw)
        print(kw, "total %.2f ms" % (dt * 1e3), {k: round(v * 1e3, 2) for k, v in T.items()})
