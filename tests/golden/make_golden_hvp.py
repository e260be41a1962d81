#!/usr/bin/env python3
"""Golden Hessian-vector products from the reference's OWN PyTorch backend (development container only; needs /root/reference).

`PytorchBackend` (BoManifolds/pymanopt_addons/tools/autodiff/_pytorch.py:35-116) is imported unmodified and compiled on costs built from the
reference's `affine_invariant_distance_torch` (Riemannian_utils/spd_utils_torch.py:53-121) and `sphere_distance_torch`
(sphere_utils_torch.py:12-55):  cost(X) = sum_ij G_ij K(X_i, Y_j), K = exp(-beta d^2) (kernels_spd.py:94-98, kernels_sphere.py:90-94),
exp(-beta d) (kernels_spd.py:185) or d itself.  Stored: the backend's `egrad(X)` and `ehess(X, U)` - the exact Hessian-vector products the reference's
trust regions use with approx_hessian=False (manifold_optimize.py:198-202) - at d = 2, 3, 5, 10, including a pair with a nearly repeated spectrum
(x2 = 2.5 x1 (1 + 1e-4 ...): at an EXACTLY repeated eigenvalue torch's double backward through symeig divides by zero and the reference returns NaN,
which is recorded as `exact_repeat_is_finite`).  torch's default dtype is float64 while the reference code runs (its eigenvalue buffer,
spd_utils_torch.py:108, is allocated in the default dtype; see make_golden_tr_traces.py).  The only shim is torch.symeig -> torch.linalg.eigh
(make_golden_tr.py).  Output: tests/golden/hvp.npz (arrays only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_tr as base  # noqa: E402,F401  (symeig shim, pymanopt stand-ins, sys.path of the reference)

from BoManifolds.pymanopt_addons.tools.autodiff._pytorch import PytorchBackend  # noqa: E402
from BoManifolds.Riemannian_utils.spd_utils_torch import affine_invariant_distance_torch  # noqa: E402
from BoManifolds.Riemannian_utils.sphere_utils_torch import sphere_distance_torch  # noqa: E402


def rand_spd(rng, n, d, lo=0.3, hi=3.0):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def rand_sym(rng, n, d):
    u = rng.standard_normal((n, d, d))
    return 0.5 * (u + u.transpose(0, 2, 1))


def flat(X):
    """identity on the values; makes the gradient autograd hands back for X CONTIGUOUS (the backward of the two reshapes copies it): the
    backend's `ehess` flattens that gradient with .view (_pytorch.py:113), which this torch's strided cholesky / bmm gradients do not allow"""
    return X.reshape(-1).reshape(X.shape)


def phi(dist, beta, mode):
    if mode == "gaussian":
        return torch.exp(-beta * dist * dist)
    if mode == "laplace":
        return torch.exp(-beta * dist)
    return dist


def main():
    torch.set_default_dtype(torch.float64)
    backend = PytorchBackend()
    out = {}
    rng = np.random.default_rng(20260930)
    dims = [2, 3, 5, 10]
    out["spd_dims"] = np.array(dims)
    out["modes"] = np.array([0, 1, 2])          # gaussian, laplace, distance
    for d in dims:
        n1, n2, beta = 3, 4, 0.7
        x1, x2 = rand_spd(rng, n1, d), rand_spd(rng, n2, d)
        # a pair with a nearly repeated spectrum of M: x2[0] = 2.5 x1[0] congruently perturbed by 1e-4
        pert = np.eye(d) + 1e-4 * np.diag(np.arange(d) / max(d - 1, 1))
        x2[0] = 2.5 * pert @ x1[0] @ pert
        x2[0] = 0.5 * (x2[0] + x2[0].T)
        G, U = rng.standard_normal((n1, n2)), rand_sym(rng, n1, d)
        out[f"spd{d}_x1"], out[f"spd{d}_x2"], out[f"spd{d}_G"], out[f"spd{d}_U"], out[f"spd{d}_beta"] = x1, x2, G, U, np.array(beta)
        for mode in ("gaussian", "laplace", "distance"):
            Gt, Yt = torch.tensor(G), torch.tensor(x2)

            # one point at a time, as the reference's solvers hand them over (a d x d array; the backend's `ehess` flattens the gradient with
            # .view, which the batched form's strided gradient does not allow)
            costs, egrads, ehesss = [], [], []
            for i in range(n1):
                def cost(X, g=Gt[i], Yt=Yt, mode=mode, beta=beta):
                    return (g * phi(affine_invariant_distance_torch(flat(X).unsqueeze(0), Yt)[0], beta, mode)).sum()
                fn = backend._compile(cost, torch.Tensor())
                costs.append(fn.cost(x1[i]))
                egrads.append(np.array(fn.egrad(x1[i]), copy=True))
                ehesss.append(np.array(fn.ehess(x1[i], U[i]), copy=True))
            out[f"spd{d}_{mode}_cost"] = np.array(sum(costs))
            out[f"spd{d}_{mode}_egrad"] = np.stack(egrads)
            out[f"spd{d}_{mode}_ehess"] = np.stack(ehesss)
            assert np.isfinite(out[f"spd{d}_{mode}_ehess"]).all()
            print(f"spd d={d} {mode}: |egrad| {np.abs(out[f'spd{d}_{mode}_egrad']).max():.3e} |ehess| {np.abs(out[f'spd{d}_{mode}_ehess']).max():.3e}", flush=True)
    # what the reference does at an exactly repeated eigenvalue (recorded, not compared)
    x1 = rand_spd(rng, 1, 3)
    fn = backend._compile(lambda X: phi(affine_invariant_distance_torch(flat(X).unsqueeze(0), torch.tensor(2.5 * x1)), 0.7, "gaussian").sum(), torch.Tensor())
    fn.egrad(x1[0])
    out["exact_repeat_is_finite"] = np.array(bool(np.isfinite(fn.ehess(x1[0], rand_sym(rng, 1, 3)[0])).all()))
    print("exactly repeated eigenvalues: reference ehess finite =", bool(out["exact_repeat_is_finite"]))

    sph = [3, 5, 10]
    out["sphere_dims"] = np.array(sph)
    for dim in sph:
        n1, n2, beta = 4, 6, 1.3
        x1 = rng.standard_normal((n1, dim))
        x1 /= np.linalg.norm(x1, axis=1, keepdims=True)
        x2 = rng.standard_normal((n2, dim))
        x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
        G, U = rng.standard_normal((n1, n2)), rng.standard_normal((n1, dim))
        Gt, Yt = torch.tensor(G), torch.tensor(x2)

        costs, egrads, ehesss = [], [], []
        for i in range(n1):
            def cost(X, g=Gt[i], Yt=Yt, beta=beta):
                dist = sphere_distance_torch(flat(X).unsqueeze(0), Yt)[0]
                return (g * torch.exp(-beta * dist * dist)).sum()
            fn = backend._compile(cost, torch.Tensor())
            costs.append(fn.cost(x1[i]))
            egrads.append(np.array(fn.egrad(x1[i]), copy=True))
            ehesss.append(np.array(fn.ehess(x1[i], U[i]), copy=True))
        out[f"sph{dim}_x1"], out[f"sph{dim}_x2"], out[f"sph{dim}_G"], out[f"sph{dim}_U"], out[f"sph{dim}_beta"] = x1, x2, G, U, np.array(beta)
        out[f"sph{dim}_cost"] = np.array(sum(costs))
        out[f"sph{dim}_egrad"] = np.stack(egrads)
        out[f"sph{dim}_ehess"] = np.stack(ehesss)
        print(f"sphere dim={dim}: |ehess| {np.abs(out[f'sph{dim}_ehess']).max():.3e}", flush=True)
    path = os.path.join(HERE, "hvp.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
