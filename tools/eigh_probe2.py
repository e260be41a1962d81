"""Development probe: the spectra of tests/test_gpu_manifold_ops.py::test_lane_group_eigen_solver_on_many_spectra, error per group."""
import sys
import numpy as np
import torch
from gabotorch_amd import _lib, ops

d = int(sys.argv[1]) if len(sys.argv) > 1 else 9
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(500 + d + 1000 * seed)
n = 200
mats = []
g = rng.standard_normal((n, d, d))
mats.append(np.einsum("nab,ncb->nac", g, g) / d + 0.1 * np.eye(d))
def with_spectrum(lam):
    q = np.linalg.qr(rng.standard_normal((lam.shape[0], d, d)))[0]
    return np.einsum("nab,nb,ncb->nac", q, lam, q)
mats.append(with_spectrum(10.0 ** rng.uniform(-4, 4, (n, d))))
lam = rng.uniform(0.5, 3.0, (n, d))
lam[:, 1] = lam[:, 0] * (1.0 + 10.0 ** rng.uniform(-12, -3, n))
mats.append(with_spectrum(lam))
lam = rng.uniform(0.5, 3.0, (n, d))
lam[:, :5] = lam[:, :1] * (1.0 + 10.0 ** rng.uniform(-9, -4, (n, 1)) * np.arange(5))
mats.append(with_spectrum(lam))
grade = 10.0 ** (-np.arange(d) * rng.uniform(0.05, 0.25, (n, 1)))
mats.append(np.einsum("na,nab,nb->nab", grade, mats[0], grade) + 1e-9 * np.eye(d))
mats.append(with_spectrum(rng.uniform(0.2, 5.0, (n, d))) * 10.0 ** rng.uniform(-6, 6, (n, 1, 1)))
mats = np.concatenate(mats)
mats = 0.5 * (mats + mats.transpose(0, 2, 1))
lam, vec = np.linalg.eigh(mats)
fun = lambda f: np.einsum("nab,nb,ncb->nac", vec, f(lam), vec)
root = ops.spd_manifold_op(_lib.GABO_SPD_SQRTM, torch.tensor(mats, device="cuda:0")).cpu().numpy()
err = np.abs(root - fun(np.sqrt)).max(axis=(1, 2)) / np.sqrt(lam.max(axis=1))
back = np.abs(np.einsum("nab,nbc->nac", root, root) - mats).max(axis=(1, 2)) / lam.max(axis=1)
bad = (back > 2e-13).sum()
print(f"d {d} seed {seed}: max back error {back.max():.2e}, matrices above 2e-13: {bad}")
for k in range(6 if "-v" in sys.argv else 0):
    sl = slice(k * n, (k + 1) * n)
    i = k * n + int(np.argmax(err[sl]))
    sp = lam[i] / lam[i].max()
    print(f"group {k}: sqrt err max {err[sl].max():.2e} (>2e-13: {(err[sl] > 2e-13).sum()}), back {back[sl].max():.2e}; worst: min rel gap {np.diff(sp).min():.2e}, cond {1 / sp.min():.1e}")
