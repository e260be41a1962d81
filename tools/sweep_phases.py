import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gabotorch_amd.manifold_optimization.manifold_optimize as mo
from tools import sweep_bench
T = {}
def wrap(name, fn):
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return inner
mo.gen_batch_initial_conditions_manifold = wrap("initial_conditions", mo.gen_batch_initial_conditions_manifold)
mo.gen_candidates_manifold = wrap("gen_candidates", mo.gen_candidates_manifold)
from gabotorch_amd.manifold_optimization import batched_trust_regions as btr
btr.BatchedTrustRegions._tcg = wrap("tcg", btr.BatchedTrustRegions._tcg)
for g in (False, True):
    sweep_bench.run_sweep("cuda:0", hip_graphs=g)
    T.clear()
    dt, *_ = sweep_bench.run_sweep("cuda:0", hip_graphs=g)
    print("graphs", g, "total %.3f" % dt, {k: round(v, 3) for k, v in T.items()})
