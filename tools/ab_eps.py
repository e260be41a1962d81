"""A/B of the QL deflation threshold: time and error of d^2 AND of d itself on near-identical pairs (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops, _lib
from tools.dev_bench import spd_set, timeit
from oracle import spd as ospd
tag = sys.argv[1]
ops.set_error_checking(False)
n, d = 4096, 10
xs = spd_set(n, d)
x = torch.tensor(xs, device="cuda")
ms = min(timeit(lambda: ops.spd_ai_pairwise(x, x, beta=0.9), iters=10, warm=3) for _ in range(3))
a = xs[:128].copy()
rng = np.random.default_rng(3)
am = ospd.vector_to_symmetric_matrix_mandel(a)
worst = {}
for scale in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6):
    pert = rng.standard_normal(am.shape) * scale
    pert = 0.5 * (pert + pert.transpose(0, 2, 1))
    bm = am @ (np.eye(d) + pert) ; bm = 0.5 * (bm + bm.transpose(0, 2, 1))
    w, v = np.linalg.eigh(bm); bm = np.einsum("nab,nb,ncb->nac", v, np.maximum(w, 1e-3), v)
    b = ospd.symmetric_matrix_to_vector_mandel(bm)
    dist = ops.spd_ai_pairwise(torch.tensor(a, device="cuda"), torch.tensor(b, device="cuda"), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    want = ospd.affine_invariant_distance(am, bm)
    dd = np.diag(dist); dw = np.diag(want)
    worst[scale] = (float(np.max(np.abs(dd - dw) / dw)), float(np.max(np.abs(dd - dw))), float(dw.mean()))
print(f"[{tag}] d=10: {ms:.3f} ms; near pairs (rel err of d, abs err of d, mean d): " + "  ".join(f"{s:g}: {v[0]:.1e} {v[1]:.1e} {v[2]:.1e}" for s, v in worst.items()))
