import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gabotorch_amd import ops
from tools.sphere_sweep_bench import run
ops.set_error_checking(False)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for _ in range(3):
    run(approx=False, constrained=False, R=R, raw=4 * R)
