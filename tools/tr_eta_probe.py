"""What do consecutive tCG steps of a restart that sits on the bound look like? (development)  Runs the config-4 sweep on the multi-launch device plan
and records, for the restarts that run to maxiter, the whitened step eta of every iteration, its ratio to the previous one and the stop reason."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gabotorch_amd import ops
from tools import sweep_bench
rec = []
orig = ops.SpdTcg.end
def end(self):
    eta, heta, stop = orig(self)
    rec.append((eta.clone(), stop.clone()))
    return eta, heta, stop
ops.SpdTcg.end = end
dt, best, val, log = sweep_bench.run_sweep("cuda:0", device_rand=True, builtin_constraint=False, native_sweep=False, device_iteration=False, maxiter=30)
it = log["per_restart_iterations"].cpu()
idx = int(torch.argsort(it, descending=True)[0])
print("restart", idx, "iterations", int(it[idx]), "records", len(rec))
prev = None
for k, (eta, stop) in enumerate(rec[:30]):
    e = eta[idx].cpu()
    line = f"it {k:2d} stop {int(stop[idx])} |eta| {float(e.norm()):.6e}"
    if prev is not None and float(prev.norm()) > 0:
        ratio = e / prev
        fin = ratio[torch.isfinite(ratio)]
        line += f"  ratio to previous: min {float(fin.min()):.17g} max {float(fin.max()):.17g}  exact quarter: {bool(torch.equal(e, 0.25 * prev))}"
    print(line)
    prev = e
