"""development: the config-4 sweep on the Python path (native_sweep=False) a few times, for rocprofv3 - python tools/sweep_once_python.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_bench import run_sweep
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for _ in range(3):
    run_sweep("cuda:0", num_restarts=R, raw_samples=4 * R, device_rand=True, builtin_constraint=True, native_sweep=False)
