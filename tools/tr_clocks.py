"""Where one trust-region iteration's cycles go (development; needs the library built by
`python tools/ab_build.py clk spd_tr_solve.hip -DGABO_TR_CLOCKS` and GABO_HIP_LIB pointing at it): runs the config-4 single-launch solve
and prints, for restart 0, the cycles between the instrumentation points of csrc/spd_tr_body.hpp / spd_acq_body.hpp."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import _lib
from tools.sweep_bench import run_sweep

lib = ctypes.CDLL(_lib.LIB_PATH)
lib.gabo_debug_clocks.restype = ctypes.c_int
lib.gabo_debug_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
run_sweep("cuda:0", num_restarts=R, device_rand=True, builtin_constraint=True, native_sweep=False)
buf = (ctypes.c_longlong * 8192)()
lib.gabo_debug_clocks(buf, 4096)                       # drop the warm-up run
MAXIT = int(sys.argv[2]) if len(sys.argv) > 2 else 6
run_sweep("cuda:0", num_restarts=R, device_rand=True, builtin_constraint=True, maxiter=MAXIT, native_sweep=False)
n = lib.gabo_debug_clocks(buf, 4096)
ev = [(int(buf[2 * k]), int(buf[2 * k + 1])) for k in range(n)]
names = {1: "iteration start", 2: "tcg_begin", 3: "builtin constraints", 4: "tcg_fd_point", 5: "acq_eval at the FD point", 6: "tcg_step",
         7: "proposal: expm, congruence, Mandel", 8: "acq_eval at the proposal", 9: "update (rho test, state)",
         100: "  acq: entry", 101: "  acq: chol + inverse of x", 102: "  acq: pairs (M, eigen, logs, F)", 103: "  acq: GP posterior + EI",
         104: "  acq: weights + accumulate", 105: "  acq: reduce"}
print(f"{n} events for restart 0")
tot = collections.OrderedDict()
prev_outer = None
prev_any = None
for tag, t in ev:
    if prev_any is not None:
        key = names.get(tag, str(tag)) if tag >= 100 else None
        if tag >= 100 and prev_any[0] >= 100 and tag != 100:
            tot.setdefault(key, []).append(t - prev_any[1])
    if tag < 100:
        if prev_outer is not None:
            tot.setdefault(names.get(tag, str(tag)), []).append(t - prev_outer[1])
        prev_outer = (tag, t)
    prev_any = (tag, t)
for k, v in tot.items():
    print(f"{k:45s} n={len(v):3d}  mean {np.mean(v):9.0f} cycles   total {np.sum(v):10.0f}")
its = [t for tag, t in ev if tag == 1]
if len(its) > 1:
    print("cycles per trust-region iteration:", [its[k + 1] - its[k] for k in range(len(its) - 1)])

# per-iteration phase table (outer tags only): which phases ran and what they cost
rows, cur, last = [], {}, None
for tag, t in ev:
    if tag >= 100:
        continue
    if tag == 1 and cur:
        rows.append(cur)
        cur = {}
    if last is not None and tag != 1:
        cur[tag] = cur.get(tag, 0) + (t - last)
    last = t
if cur:
    rows.append(cur)
print("iteration: " + "  ".join(f"{names[k][:14]:>14s}" for k in range(2, 10)))
for k, r in enumerate(rows[:40]):
    print(f"{k:9d}: " + "  ".join(f"{r.get(tag, 0):14d}" for tag in range(2, 10)) + f"   total {sum(r.values())}")
