import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools.sphere_sweep_bench import run
for kw in (dict(approx=False, constrained=False), dict(approx=True, constrained=False)):
    run(**kw)
    dt, val, its, log = run(**kw)
    it = log["per_restart_iterations"]
    print(kw, f"{dt*1e3:.2f} ms iterations mean {float(it.float().mean()):.1f} max {int(it.max())} at maxiter {int((it >= 50).sum())}")
