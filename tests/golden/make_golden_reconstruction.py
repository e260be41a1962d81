#!/usr/bin/env python3
"""Golden vectors for the nested-SPD reconstruction costs (development container only; needs /root/reference).

Imports the reference's nested_spd_optimization.py UNMODIFIED.  Its module-level imports of gpytorch / pymanopt (absent here)
are satisfied by empty stand-in modules - the two cost functions called below use neither.  Outputs: cost values and autograd
gradients with respect to the complement basis V, the bottom block C and the contraction K, plus logm / sqrtm gradients."""
import collections
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("GABO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
_R = collections.namedtuple("symeig", ["eigenvalues", "eigenvectors"])
torch.symeig = lambda A, eigenvectors=False, upper=True: _R(*torch.linalg.eigh(A, UPLO="U" if upper else "L"))
sys.path.insert(0, REF)
for name in ("gpytorch", "pymanopt", "pymanopt.manifolds", "pymanopt.solvers", "pymanopt.solvers.solver"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["pymanopt.solvers.solver"].Solver = object
sys.modules["pymanopt"].manifolds = sys.modules["pymanopt.manifolds"]
sys.modules["pymanopt"].solvers = sys.modules["pymanopt.solvers"]
sys.modules["pymanopt.solvers"].solver = sys.modules["pymanopt.solvers.solver"]

from BoManifolds.nested_mappings import nested_spd_optimization as nso  # noqa: E402
from BoManifolds.Riemannian_utils import spd_utils_torch as sut  # noqa: E402
from BoManifolds.nested_mappings import nested_spheres_optimization as nsso  # noqa: E402
from BoManifolds.nested_mappings import nested_spheres_utils as nsu  # noqa: E402
from BoManifolds.nested_mappings import nested_spd_constraints_utils as nscu  # noqa: E402


def rand_spd(rng, n, d, lo=0.3, hi=3.0):
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return 0.5 * (m + m.transpose(0, 2, 1))


def main():
    out = {}
    rng = np.random.default_rng(61)
    for tag, D, d, N in (("a", 4, 2, 6), ("b", 5, 2, 5)):
        X = rand_spd(rng, N, D)
        R = np.linalg.qr(rng.standard_normal((D, D)))[0]
        W, V = R[:, :d], R[:, d:]
        Y = np.einsum("da,ndc,cb->nab", W, X, W)
        C = rand_spd(rng, 1, D - d)[0]
        K = rng.standard_normal((d, D - d))
        K = 0.6 * K / np.linalg.norm(K)
        out.update({f"{tag}_X": X, f"{tag}_W": W, f"{tag}_V": V, f"{tag}_Y": Y, f"{tag}_C": C, f"{tag}_K": K})
        for name, fn in (("ai", nso.min_affine_invariant_distance_reconstruction_cost), ("le", nso.min_log_euclidean_distance_reconstruction_cost)):
            tV, tC, tK = (torch.tensor(a, requires_grad=True) for a in (V, C, K))
            cost = fn(torch.tensor(X), torch.tensor(Y), torch.tensor(W), tV, tC, tK)
            cost.backward()
            out[f"{tag}_{name}_cost"] = np.array(cost.item())
            out[f"{tag}_{name}_gV"], out[f"{tag}_{name}_gC"], out[f"{tag}_{name}_gK"] = tV.grad.numpy(), tC.grad.numpy(), tK.grad.numpy()
    # matrix functions and their autograd gradients
    A = rand_spd(rng, 5, 4)
    G = rng.standard_normal((5, 4, 4))
    for name, fn in (("logm", sut.logm_torch), ("sqrtm", sut.sqrtm_torch)):
        grads, vals = [], []
        for k in range(5):
            a = torch.tensor(A[k], requires_grad=True)
            y = fn(a)
            (y * torch.tensor(G[k])).sum().backward()
            vals.append(y.detach().numpy())
            grads.append(a.grad.numpy())
        out[f"{name}_val"], out[f"{name}_grad"] = np.stack(vals), np.stack(grads)
    out["mf_A"], out["mf_G"] = A, G
    x1, x2 = rand_spd(rng, 4, 3), rand_spd(rng, 5, 3)
    out["frob_x1"], out["frob_x2"] = x1, x2
    out["frob_d"] = sut.frobenius_distance_torch(torch.tensor(x1), torch.tensor(x2)).numpy()
    # eigenvalue constraints stated in the original space (nested_spd_constraints_utils.py:14-73), value and gradient w.r.t. the nested point
    for k in range(3):
        y = torch.tensor(out["a_Y"][k], requires_grad=True)
        args = [torch.tensor(out[f"a_{n}"]) for n in ("W", "V", "C", "K")]
        fmax = nscu.max_eigenvalue_nested_spd_constraint(y, 4.0, *args)
        fmax.backward()
        out[f"nc_max{k}"], out[f"nc_gmax{k}"] = np.array(fmax.item()), y.grad.numpy().copy()
        y2 = torch.tensor(out["a_Y"][k], requires_grad=True)
        fmin = nscu.min_eigenvalue_nested_spd_constraint(y2, 0.1, *args)
        fmin.backward()
        out[f"nc_min{k}"], out[f"nc_gmin{k}"] = np.array(fmin.item()), y2.grad.numpy().copy()
    # nested spheres: reconstruction error as a function of the distances to the axes (nested_spheres_optimization.py:20-38)
    dim, latent, n = 5, 3, 9
    xs = rng.standard_normal((n, dim)); xs /= np.linalg.norm(xs, axis=1, keepdims=True)
    axes = []
    for dd in range(dim, latent, -1):
        a = rng.standard_normal((1, dd)); a /= np.linalg.norm(a)
        axes.append(a)
    r_proj = [torch.tensor([[1.2]], dtype=torch.float64), torch.tensor([[1.4]], dtype=torch.float64)]
    sub = nsu.projection_from_sphere_to_subsphere(torch.tensor(xs), [torch.tensor(a) for a in axes], r_proj)[-1]
    r_eval = [torch.tensor([[1.0]], dtype=torch.float64, requires_grad=True), torch.tensor([[1.5]], dtype=torch.float64, requires_grad=True)]
    cost = nsso.min_error_reconstruction_cost(torch.tensor(xs), sub, [torch.tensor(a) for a in axes], r_eval)
    cost.backward()
    out.update({"ns_x": xs, "ns_sub": sub.numpy(), "ns_axis0": axes[0], "ns_axis1": axes[1], "ns_r": np.array([1.0, 1.5]),
                "ns_cost": np.array(cost.item()), "ns_grad": np.array([r_eval[0].grad.item(), r_eval[1].grad.item()])})
    np.savez_compressed(os.path.join(HERE, "reconstruction.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith("cost")}, out["a_ai_cost"], out["a_le_cost"])


if __name__ == "__main__":
    main()
