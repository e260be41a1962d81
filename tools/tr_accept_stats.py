"""How many of the long-running restarts' trust-region iterations are REJECTED proposals?  (development: decides whether evaluating the
proposal's acquisition VALUE first and its gradient only on acceptance can pay.)  Runs the config-4 sweep on the two-launch plan
(propose / update) and compares every restart's iterate before and after each update."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gabotorch_amd import ops
from tools import sweep_bench

acc = {}
orig = ops.SpdTr.update
def update(self, x, fx, g, ng, Delta, active, iters, *a, **k):
    before, act = x.clone(), active.clone()
    orig(self, x, fx, g, ng, Delta, active, iters, *a, **k)
    moved = (before != x).flatten(1).any(1)
    st = acc.setdefault("s", {"acc": torch.zeros(x.shape[0], dtype=torch.long, device=x.device), "rej": torch.zeros(x.shape[0], dtype=torch.long, device=x.device)})
    acc["delta"] = Delta
    st["acc"] += (moved & (act != 0)).long()
    st["rej"] += (~moved & (act != 0)).long()
ops.SpdTr.update = update
dt, best, val, log = sweep_bench.run_sweep("cuda:0", device_rand=True, builtin_constraint=True, native_sweep=False, device_solve=False)
it = log["per_restart_iterations"].cpu()
a, r = acc["s"]["acc"].cpu(), acc["s"]["rej"].cpu()
print("restarts", len(it), "iterations: max", int(it.max()), "mean %.1f" % float(it.float().mean()), "restarts at maxiter:", int((it >= 100).sum()))
long = it >= int(it.max())
print("long-running restarts: accepted per restart mean %.1f, rejected mean %.1f" % (float(a[long].float().mean()), float(r[long].float().mean())))
print("all restarts: accepted total", int(a.sum()), "rejected total", int(r.sum()))
top = torch.argsort(it, descending=True)[:12]
print("final trust radius of the top restarts:", [float(acc["delta"][i]) for i in top[:6]], " (Delta0 = sqrt(15)/8 = 0.484; a rejection quarters it)")
print("final gradient norms:", [float(log["final_gradnorm"][i]) for i in top[:6]])
print("top restarts (index, iters, accepted, rejected):", [(int(i), int(it[i]), int(a[i]), int(r[i])) for i in top])
