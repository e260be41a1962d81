"""Bit-identity of the single-launch solve across builds (development): runs the config-4 sweep (Python path, device sampler) and writes the
final iterates, costs and iteration counts of all restarts to a file; run once per library (GABO_HIP_LIB) and compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.sweep_bench import run_sweep
out = sys.argv[1]
res = {}
for tag, kw in (("c4", dict()), ("c4_strict", dict(strict=True)), ("c4_free", dict(constraint=False)), ("c4_1000", dict(maxiter=300))):
    dt, best, val, log = run_sweep("cuda:0", device_rand=True, builtin_constraint=True, native_sweep=False, **kw)
    res[tag + "_best"] = best.cpu().numpy()
    res[tag + "_cost"] = log["final_cost"].cpu().numpy()
    res[tag + "_iters"] = log["per_restart_iterations"].cpu().numpy()
    res[tag + "_ms"] = np.array(dt * 1e3)
    print(tag, "ms %.3f" % (dt * 1e3), "max iters", int(res[tag + "_iters"].max()))
np.savez(out, **res)
