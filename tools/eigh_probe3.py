"""Development probe: block-diagonal inputs (an exactly zero interior off-diagonal of the tridiagonal form) through sqrtm."""
import numpy as np
import torch
from gabotorch_amd import _lib, ops

rng = np.random.default_rng(3)
for d in (4, 6, 8, 10, 12, 16, 20, 28, 32):
    n = 64
    k = d // 2
    g1 = rng.standard_normal((n, k, k)); g2 = rng.standard_normal((n, d - k, d - k))
    m = np.zeros((n, d, d))
    m[:, :k, :k] = np.einsum("nab,ncb->nac", g1, g1) / k + 0.1 * np.eye(k)
    m[:, k:, k:] = np.einsum("nab,ncb->nac", g2, g2) / (d - k) + 0.1 * np.eye(d - k)
    root = ops.spd_manifold_op(_lib.GABO_SPD_SQRTM, torch.tensor(m, device="cuda:0")).cpu().numpy()
    back = np.abs(np.einsum("nab,nbc->nac", root, root) - m).max(axis=(1, 2)) / np.linalg.eigvalsh(m).max(axis=1)
    print(f"d {d:3d}: block-diagonal sqrtm(A)^2 - A: max {back.max():.2e}, median {np.median(back):.2e}")
