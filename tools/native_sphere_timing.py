"""sphere sweep S^9, 512 restarts: native driver against the Python path (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sphere_sweep_bench import run
for approx in (False, True):
    for native in (True, False):
        for _ in range(3):
            run(approx=approx, constrained=False, native=native)
        ts = sorted(run(approx=approx, constrained=False, native=native)[0] for _ in range(9))
        print(f"exact Hessian={not approx} native={native}: min {ts[0]*1e3:.3f} ms, median {ts[4]*1e3:.3f} ms")
