"""Host-side phase timing of the graph-replay sweep (warm): where the remaining milliseconds go."""
import os, sys, time, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.sweep_bench import run_sweep
from gabotorch_amd.manifold_optimization import batched_trust_regions as btr, manifold_optimize as mo

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)

def wrap(obj, name, label, static=False):
    fn = getattr(obj, name)
    def inner(*a, **k):
        if static:
            a = a[1:] if a and isinstance(a[0], btr.BatchedTrustRegions) else a
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); acc[label] += time.perf_counter() - t0; cnt[label] += 1
        return out
    setattr(obj, name, inner)

for _ in range(2):
    run_sweep("cuda:0", hip_graphs=True, batched_rand=True)
wrap(mo, "gen_batch_initial_conditions_manifold", "init_conditions")
wrap(btr.BatchedTrustRegions, "_constraint_values_grads", "constraints", static=True)
wrap(btr.BatchedTrustRegions, "solve", "solve_total")
wrap(mo, "gen_candidates_manifold", "gen_candidates_total")
wrap(btr.BatchedTrustRegions, "_solve_device", "solve_device")
wrap(torch.cuda.CUDAGraph, "replay", "graph_replay")
wrap(mo.FusedAcquisition, "build", "fused_build")
import gabotorch_amd.models as models
wrap(models.ExactGP, "_train_cache", "gp_train_cache")
dt, *_ = run_sweep("cuda:0", hip_graphs=True, batched_rand=True)
print(json.dumps({"sweep_s_with_syncs": dt, "phases_ms": {k: round(v * 1e3, 3) for k, v in acc.items()}, "calls": cnt}, indent=1))
