"""development: per-restart cycles of the two-wave solve kernel (library built with -DGABO_DUO_TIMES: tools/ab_build.py times spd_tr_solve_duo.hip -DGABO_DUO_TIMES ...)
    GABO_HIP_LIB=gabotorch_amd/libgabo_hip_times.so python tools/duo_times.py [R]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.sweep_bench import run_sweep
from gabotorch_amd import _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kw = dict(num_restarts=R, raw_samples=4 * R, device_rand=True, builtin_constraint=True)
for _ in range(3):
    run_sweep("cuda:0", **kw)
buf = (ctypes.c_longlong * (4 * R))()
_lib.load().gabo_debug_duo_times(buf, R)
a = np.array(buf[:]).reshape(R, 4)
order = np.argsort(-a[:, 0])
print("restart  cycles  iterations  hits  misses   cycles/iteration")
for k in order[:12]:
    print(f"{k:6d} {a[k, 0]:9d} {a[k, 1]:6d} {a[k, 2]:6d} {a[k, 3]:6d}   {a[k, 0] / max(1, a[k, 1]):10.0f}")
print("...")
for k in order[-4:]:
    print(f"{k:6d} {a[k, 0]:9d} {a[k, 1]:6d} {a[k, 2]:6d} {a[k, 3]:6d}   {a[k, 0] / max(1, a[k, 1]):10.0f}")
# the (tag, cycle) pairs both waves of restart GABO_DUO_CLOCKS_BLOCK (default 55) recorded during the LAST sweep
try:
    lib = _lib.load()
    lib.gabo_debug_duo_clocks((ctypes.c_longlong * 2)(), 0, 1)
    run_sweep("cuda:0", **kw)
    cb = (ctypes.c_longlong * (2 * 8192))()
    n = lib.gabo_debug_duo_clocks(cb, 8192, 1)
    pairs = np.array(cb[:2 * n]).reshape(n, 2)
    t0 = pairs[:, 1].min()
    rows = {0: [], 1: []}
    for tag, c in pairs:
        rows[int(tag) // 1000].append((int(tag) % 1000, int(c - t0)))
    for wv in (0, 1):
        rows[wv].sort(key=lambda r: r[1])
        print(f"wave {wv}: " + " ".join(f"{tag}@{c}" for tag, c in rows[wv][:150]))
except Exception as e:
    print("no clocks:", e)
