"""Development probe: error of sqrtm against numpy's eigh for pairs of eigenvalues at a given relative gap (the lane-group solver of wave_eigh.hpp)."""
import sys
import numpy as np
import torch
from gabotorch_amd import _lib, ops

d = int(sys.argv[1]) if len(sys.argv) > 1 else 9
rng = np.random.default_rng(500 + d)
n = 400
lam = rng.uniform(0.5, 3.0, (n, d))
gaps = 10.0 ** rng.uniform(-12, -3, n)
lam[:, 1] = lam[:, 0] * (1.0 + gaps)
q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
mats = np.einsum("nab,nb,ncb->nac", q, lam, q)
mats = 0.5 * (mats + mats.transpose(0, 2, 1))
l, v = np.linalg.eigh(mats)
want = np.einsum("nab,nb,ncb->nac", v, np.sqrt(l), v)
got = ops.spd_manifold_op(_lib.GABO_SPD_SQRTM, torch.tensor(mats, device="cuda:0")).cpu().numpy()
err = np.abs(got - want).max(axis=(1, 2)) / np.sqrt(l.max(axis=1))
order = np.argsort(-err)[:12]
for i in order:
    others = np.sort(lam[i])
    print(f"err {err[i]:.2e}  pair gap {gaps[i]:.2e}  smallest other gap {np.min(np.diff(others)[np.diff(others) > 2 * gaps[i] * 3]):.2e}")
print("median", np.median(err))
