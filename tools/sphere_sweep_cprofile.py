"""development: cProfile of the S^9 exact-Hessian sweep (host side) - python tools/sphere_sweep_cprofile.py [R]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import ops
from tools.sphere_sweep_bench import run
ops.set_error_checking(False)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kw = dict(approx=False, constrained=False, R=R, raw=4 * R)
for _ in range(5):
    run(**kw)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    run(**kw)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(30)
st.sort_stats("tottime").print_stats(14)
