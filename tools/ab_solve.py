"""A/B of the single-launch trust-region solve (development): config-4 sweep, 512 restarts, constraint as functools.partial of the built-in.
Usage: GABO_HIP_LIB=... python tools/ab_solve.py tag [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_bench import run_sweep
tag = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
run_sweep("cuda:0", num_restarts=R, batched_rand=True, builtin_constraint=True)
for name, kw in (("host samples", dict(batched_rand=True)), ("device samples", dict(device_rand=True))):
    ts = []
    for _ in range(6):
        dt, best, val, log = run_sweep("cuda:0", num_restarts=R, builtin_constraint=True, **kw)
        ts.append(dt)
    print(f"[{tag}] R={R} single-launch solve, {name}: min {min(ts)*1e3:.2f} ms  median {sorted(ts)[len(ts)//2]*1e3:.2f} ms  EI*={val:.12e}  TR iterations={log['iterations']}")
