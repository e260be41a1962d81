"""Small-d SPD Gram (the latent spaces of the nested kernels): time and worst error against the oracle, including near-identical and
ill-conditioned pairs (development).  Usage: GABO_HIP_LIB=... python tools/ab_small_d.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops, _lib
from tools.dev_bench import spd_set, timeit
from oracle import spd as ospd

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(_lib.LIB_PATH)
ops.set_error_checking(False)
n, beta = 4096, 0.6 + float(np.log(2.0))
for d in (2, 3, 4):
    xs = spd_set(n, d)
    x = torch.tensor(xs, device="cuda")
    ms = min(timeit(lambda: ops.spd_ai_pairwise(x, x, beta=beta), iters=20, warm=5) for _ in range(3))
    a, b = xs[:192].copy(), xs[3000:3384].copy()
    # near-identical pairs and badly conditioned matrices in the checked block
    b[:64] = a[:64] * (1.0 + 1e-9 * np.arange(64))[:, None]
    b[64:96] = a[64:96]
    rng = np.random.default_rng(d)
    q = np.linalg.qr(rng.standard_normal((32, d, d)))[0]
    lam = np.exp(rng.uniform(-9, 9, (32, d)))
    ill = np.einsum("nab,nb,ncb->nac", q, lam, q); ill = 0.5 * (ill + ill.transpose(0, 2, 1))
    b[96:128] = ospd.symmetric_matrix_to_vector_mandel(ill)
    dist = ops.spd_ai_pairwise(torch.tensor(a, device="cuda"), torch.tensor(b, device="cuda"), mode=_lib.GABO_OUT_DISTANCE).cpu().numpy()
    kk = ops.spd_ai_pairwise(torch.tensor(a, device="cuda"), torch.tensor(b, device="cuda"), beta=beta).cpu().numpy()
    want = ospd.affine_invariant_distance(ospd.vector_to_symmetric_matrix_mandel(a), ospd.vector_to_symmetric_matrix_mandel(b))
    err = float(np.max(np.abs(dist ** 2 - want ** 2) / want ** 2))
    errk = float(np.max(np.abs(kk - np.exp(-beta * want ** 2)) / np.exp(-beta * want ** 2)))
    print(f"[{tag}] SPD d={d} N={n}: {ms*1e3:.1f} us  {n*n/ms*1e3:.3e} pairs/s   max rel err of d^2 {err:.2e}, of K {errk:.2e}, d(x,x) = {dist[64,64]:.3e}")
