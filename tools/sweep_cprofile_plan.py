"""development: cProfile of joint_optimize_manifold alone (config-4 native sweep, 64 restarts), sorted by own time - python tools/sweep_cprofile_plan.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import sweep_bench
from gabotorch_amd.manifold_optimization import manifold_optimize as mo
pr = cProfile.Profile()
orig = mo.joint_optimize_manifold


def wrapped(*a, **k):
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()


kw = dict(num_restarts=64, raw_samples=256, builtin_constraint=True, device_rand=True)
for _ in range(5):
    sweep_bench.run_sweep("cuda:0", **kw)
sweep_bench.joint_optimize_manifold = wrapped
for _ in range(50):
    sweep_bench.run_sweep("cuda:0", **kw)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
