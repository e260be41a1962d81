"""cProfile of the config-4 sweep's HOST side (development): where the ~1.4 ms around the 2.9-ms solve kernel go."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import sweep_bench
kw = dict(builtin_constraint=True, device_rand=True, native_sweep="--python" not in sys.argv)
for _ in range(5):
    sweep_bench.run_sweep("cuda:0", **kw)
pr = cProfile.Profile()
times = []
for _ in range(20):
    pr.enable()
    times.append(sweep_bench.run_sweep("cuda:0", **kw)[0])
    pr.disable()
print("sweep ms (timed inside run_sweep, joint_optimize_manifold only):", [round(t * 1e3, 2) for t in times])
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
