"""A/B of the QL deflation threshold for GAUSSIAN values without a distance output (round 5): time of the headline Gram and the error of K on the
benchmark set and on nearly identical pairs (development).  GABO_HIP_LIB selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
from tools.dev_bench import spd_set
from oracle import spd as ospd
tag = sys.argv[1]
ops.set_error_checking(False)
beta = 0.2 + float(np.log(2.0))
out = []
for d in (10, 7, 5):
    n = 4096
    xs = spd_set(n, d)
    x = torch.tensor(xs, device="cuda")
    for _ in range(30):
        ops.spd_ai_pairwise(x, x, beta=beta)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    ev[0].record()
    for k in range(30):
        kk = ops.spd_ai_pairwise(x, x, beta=beta)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(30)]))
    want = ospd.spd_ai_gaussian_kernel(xs[:192], xs[:256], beta)
    got = kk[:192, :256].cpu().numpy()
    err = float(np.max(np.abs(got - want) / np.abs(want)))
    # nearly identical pairs: K -> 1
    a = xs[:128].copy()
    am = ospd.vector_to_symmetric_matrix_mandel(a)
    rng = np.random.default_rng(3)
    worst = []
    for scale in (1e-2, 1e-4, 1e-6):
        pert = rng.standard_normal(am.shape) * scale
        pert = 0.5 * (pert + pert.transpose(0, 2, 1))
        bm = am @ (np.eye(d) + pert)
        bm = 0.5 * (bm + bm.transpose(0, 2, 1))
        b = ospd.symmetric_matrix_to_vector_mandel(bm)
        g2 = ops.spd_ai_pairwise(torch.tensor(a, device="cuda"), torch.tensor(b, device="cuda"), beta=beta).cpu().numpy()
        w2 = ospd.spd_ai_gaussian_kernel(a, b, beta)
        worst.append(float(np.max(np.abs(g2 - w2) / np.abs(w2))))
    out.append(f"d={d}: {ms:.4f} ms, max rel err of K: benchmark block {err:.1e}, near pairs (1e-2, 1e-4, 1e-6) " + " ".join(f"{v:.1e}" for v in worst))
print(f"[{tag}] " + " | ".join(out))
