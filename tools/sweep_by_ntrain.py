"""The config-4 sweep at the training-set sizes of a GaBO run's first iterations (development): time, iterations, restarts at maxiter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.sweep_bench import run_sweep
for n in (5, 8, 12, 20, 35, 50):
    for seed in (1234, 7):
        kw = dict(device_rand=True, builtin_constraint=True, n_train=n, seed=seed)
        run_sweep("cuda:0", **kw)
        ts = sorted(run_sweep("cuda:0", **kw)[0] for _ in range(3))
        os.environ["GABO_TR_NO_SHORTCUTS"] = "1"
        t0 = sorted(run_sweep("cuda:0", **kw)[0] for _ in range(3))[0]
        del os.environ["GABO_TR_NO_SHORTCUTS"]
        log = run_sweep("cuda:0", **kw)[3]
        it = log["per_restart_iterations"]
        print(f"n_train {n:3d} seed {seed}: sweep {ts[0]*1e3:6.2f} ms (without the solve shortcuts {t0*1e3:6.2f}); iterations mean {float(it.float().mean()):5.1f} max {int(it.max())}, restarts at maxiter {int((it >= 100).sum())}")
