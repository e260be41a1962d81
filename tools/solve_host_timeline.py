"""Host-side timeline of BatchedTrustRegions._solve_device on the one-launch path (development): wall-clock of every statement group
WITHOUT device synchronisation in between (what the host spends enqueueing), then the wait for the solve kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gabotorch_amd import ops
from gabotorch_amd.manifold_optimization import batched_trust_regions as btr
from tools import sweep_bench

marks = []
def mark(name):
    marks.append((name, time.perf_counter()))

orig_cost_egrad = None
def patch(fused_cls):
    global orig_cost_egrad
    orig_cost_egrad = fused_cls.cost_egrad
    def ce(self, x):
        mark("-> cost_egrad")
        r = orig_cost_egrad(self, x)
        mark("<- cost_egrad")
        return r
    fused_cls.cost_egrad = ce

from gabotorch_amd.fused_acquisition import FusedAcquisition
patch(FusedAcquisition)
from gabotorch_amd import manifolds
for nm in ("egrad2rgrad", "norm"):
    f = getattr(manifolds.PositiveDefinite, nm)
    def mk(f, nm):
        def w(*a, **k):
            mark("-> " + nm); r = f(*a, **k); mark("<- " + nm); return r
        return staticmethod(w)
    setattr(manifolds.PositiveDefinite, nm, mk(f, nm))
oTr = ops.SpdTr.__init__
def tr_init(self, *a, **k):
    mark("-> SpdTr()"); oTr(self, *a, **k); mark("<- SpdTr()")
ops.SpdTr.__init__ = tr_init
oS = ops.SpdTr.solve
def tr_solve(self, *a, **k):
    mark("-> TR.solve launch"); r = oS(self, *a, **k); mark("<- TR.solve launch"); return r
ops.SpdTr.solve = tr_solve
oSolve = btr.BatchedTrustRegions._solve_device
def sd(self, *a, **k):
    mark("-> _solve_device"); r = oSolve(self, *a, **k); mark("<- _solve_device (after the iters.max().item() wait)"); return r
btr.BatchedTrustRegions._solve_device = sd
oSolveTop = btr.BatchedTrustRegions.solve
def st(self, *a, **k):
    torch.cuda.synchronize(); mark("-> solve"); r = oSolveTop(self, *a, **k); mark("<- solve"); return r
btr.BatchedTrustRegions.solve = st

for _ in range(5):
    sweep_bench.run_sweep("cuda:0", builtin_constraint=True, device_rand=True)
for rep in range(3):
    marks.clear()
    sweep_bench.run_sweep("cuda:0", builtin_constraint=True, device_rand=True)
    t0 = marks[0][1]
    print("--- sweep", rep)
    prev = t0
    for name, t in marks:
        print(f"  {1e3 * (t - t0):7.3f} ms  (+{1e3 * (t - prev):6.3f})  {name}")
        prev = t
