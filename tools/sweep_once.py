"""development: the config-4 single-launch sweep a few times (for rocprofv3 / tools/pmc.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_bench import run_sweep
for _ in range(3):
    run_sweep("cuda:0", num_restarts=512, device_rand=True, builtin_constraint=True)
