"""config-4 sweep, device-drawn raw samples: the native driver (gabo_spd_sweep_score / _solve) against the Python path (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_bench import run_sweep
for native in (True, False, True, False):
    for _ in range(3):
        run_sweep("cuda:0", device_rand=True, builtin_constraint=True, native_sweep=native)
    ts = sorted(run_sweep("cuda:0", device_rand=True, builtin_constraint=True, native_sweep=native)[0] for _ in range(9))
    print(f"native={native}: min {ts[0]*1e3:.3f} ms, median {ts[4]*1e3:.3f} ms")
