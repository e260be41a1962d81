#!/usr/bin/env python3
"""Randomised parity soak: pairwise kernels, their gradients and the manifold operations against the numpy oracle on random shapes.
    python tests/soak_parity.py [--cases 150] [--seed 0]     (a test driver, not collected by pytest: long randomised runs)"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _lib, ops                       # noqa: E402
from oracle import sphere as osph                         # noqa: E402
from oracle import spd as ospd                            # noqa: E402


def rand_spd(rng, shape, d, lo=0.05, hi=5.0):
    n = int(np.prod(shape))
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(lo, hi, (n, d)), q)
    return (0.5 * (m + m.transpose(0, 2, 1))).reshape(*shape, d, d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = "cuda:0"
    worst = {}

    def note(tag, err):
        worst[tag] = max(worst.get(tag, 0.0), float(err))

    for case in range(a.cases):
        d = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16]))
        n1, n2 = int(rng.integers(1, 70)), int(rng.integers(1, 140))
        batch = () if rng.random() < 0.6 else (int(rng.integers(1, 4)),)
        A, B = rand_spd(rng, batch + (n1,), d), rand_spd(rng, batch + (n2,), d)
        x1, x2 = ospd.symmetric_matrix_to_vector_mandel(A), ospd.symmetric_matrix_to_vector_mandel(B)
        beta = float(rng.uniform(0.1, 1.5))
        mode = int(rng.choice([_lib.GABO_OUT_GAUSSIAN, _lib.GABO_OUT_LAPLACE, _lib.GABO_OUT_DISTANCE]))
        t1, t2 = torch.tensor(x1, device=dev), torch.tensor(x2, device=dev)
        got = ops.spd_ai_pairwise(t1, t2, beta, mode).cpu().numpy()
        dist = ospd.affine_invariant_distance(A, B)
        want = dist if mode == _lib.GABO_OUT_DISTANCE else np.exp(-beta * (dist ** 2 if mode == _lib.GABO_OUT_GAUSSIAN else dist))
        note(f"spd_fwd_mode{mode}", np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
        if d <= 12 and rng.random() < 0.5:
            go = rng.standard_normal(got.shape)
            g = ops.spd_ai_backward(t1, t2, torch.tensor(go, device=dev), beta, mode, wrt=1).cpu().numpy()
            h = 1e-6
            k = tuple(int(rng.integers(0, s)) for s in x1.shape)
            xp, xm = x1.copy(), x1.copy()
            xp[k] += h
            xm[k] -= h
            fp = ops.spd_ai_pairwise(torch.tensor(xp, device=dev), t2, beta, mode).cpu().numpy()
            fm = ops.spd_ai_pairwise(torch.tensor(xm, device=dev), t2, beta, mode).cpu().numpy()
            fd = ((fp - fm) * go).sum() / (2 * h)
            note("spd_bwd_fd", abs(g[k] - fd) / max(1.0, abs(fd)))
        # sphere
        dim = int(rng.integers(2, 60))
        s1 = rng.standard_normal(batch + (n1, dim))
        s2 = rng.standard_normal(batch + (n2, dim))
        s1 /= np.linalg.norm(s1, axis=-1, keepdims=True)
        s2 /= np.linalg.norm(s2, axis=-1, keepdims=True)
        if rng.random() < 0.3:
            s2[..., 0, :] = s1[..., 0, :]              # an identical pair: the clamp edge
        gs = ops.sphere_pairwise(torch.tensor(s1, device=dev), torch.tensor(s2, device=dev), beta, _lib.GABO_OUT_GAUSSIAN).cpu().numpy()
        note("sphere_fwd", np.max(np.abs(gs - osph.sphere_gaussian_kernel(s1, s2, beta))))
        # log-Euclidean kernel (O(N) matrix logarithms + Frobenius pairs) and its gradient
        if d <= 12:
            from gabotorch_amd.kernel_utils.kernels_spd import SpdLogEuclideanGaussianKernel
            ls = float(rng.uniform(0.7, 2.0))
            kern = SpdLogEuclideanGaussianKernel().double()
            kern.lengthscale = torch.tensor(ls, dtype=torch.float64)
            q1 = t1.clone().requires_grad_(True)
            kle = kern.forward(q1, t2)
            note("le_fwd", np.max(np.abs(kle.detach().cpu().numpy() - ospd.log_euclidean_gaussian_kernel(x1, x2, ls))))
            go = rng.standard_normal(kle.shape)
            (kle * torch.tensor(go, device=dev)).sum().backward()
            g1 = ospd.log_euclidean_gaussian_kernel_grads(x1, x2, ls, go)[0]
            note("le_bwd", np.max(np.abs(q1.grad.cpu().numpy() - g1)) / max(1.0, np.abs(g1).max()))
        # marginal likelihood kernel on a random distance matrix
        from oracle import gp as ogp
        nm = int(rng.integers(1, 161))
        pts = rng.standard_normal((nm, 3))
        e = ((pts[:, None] - pts[None]) ** 2).sum(-1)
        yv = rng.standard_normal(nm)
        par = (float(rng.uniform(0.2, 1.5)), float(rng.uniform(0.5, 2.0)), float(rng.uniform(1e-3, 0.2)), float(rng.normal()))
        ll, gl = ogp.marginal_log_likelihood(e, yv, *par)
        outm = ops.gp_mll(torch.tensor(e, device=dev), torch.tensor(yv, device=dev), *par)
        note("gp_mll_value", abs(outm[0] - ll) / max(1.0, abs(ll)))
        note("gp_mll_grad", np.max(np.abs(np.array(outm[1:5]) - gl)) / max(1.0, np.abs(gl).max()))
        # manifold ops: logm / expm round trip and log / exp maps
        M = torch.tensor(rand_spd(rng, (5,), d), device=dev)
        lg = ops.spd_manifold_op(_lib.GABO_SPD_LOGM, M)
        note("expm_logm_roundtrip", (ops.spd_manifold_op(_lib.GABO_SPD_EXPM, lg) - M).abs().max() / M.abs().max())
        Y = torch.tensor(rand_spd(rng, (5,), d), device=dev)
        U = ops.spd_manifold_op(_lib.GABO_SPD_LOG, M, Y)
        note("exp_log_map_roundtrip", (ops.spd_manifold_op(_lib.GABO_SPD_EXP, M, U) - Y).abs().max() / Y.abs().max())
        # nested-sphere chain: every level of the projection and of the lift, reconstruction cost, its gradient by finite differences,
        # and the nested SPD reconstruction cost
        if case % 3 == 0:
            from gabotorch_amd.nested_mappings.nested_spheres_utils import projection_from_sphere_to_subsphere, projection_from_subsphere_to_sphere
            Dn = int(rng.integers(3, 90))
            lat = int(rng.integers(2, min(Dn, 8)))
            L = Dn - lat
            npts = int(rng.integers(1, 40))
            axes_np = []
            for k in range(L):
                ax = rng.standard_normal(Dn - k)
                axes_np.append(ax / np.linalg.norm(ax))
            r_np = rng.uniform(0.4, 2.7, L)
            xs = rng.standard_normal((npts, Dn))
            xs /= np.linalg.norm(xs, axis=1, keepdims=True)
            axes_t = [torch.tensor(ax, device=dev) for ax in axes_np]
            dists_t = [torch.tensor([[v]], dtype=torch.float64) for v in r_np]
            down = projection_from_sphere_to_subsphere(torch.tensor(xs, device=dev), axes_t, dists_t)
            want_down = osph.projection_from_sphere_to_subsphere(xs, axes_np, r_np)
            note("nested_sphere_project", max(np.max(np.abs(g_.cpu().numpy() - w_)) for g_, w_ in zip(down, want_down)))
            zs = want_down[-1]
            up = projection_from_subsphere_to_sphere(torch.tensor(zs, device=dev), axes_t, dists_t)
            note("nested_sphere_lift", max(np.max(np.abs(g_.cpu().numpy() - w_)) for g_, w_ in zip(up, osph.projection_from_subsphere_to_sphere(zs, axes_np, r_np))))
            rec = ops.NestedSphereReconstruction(torch.tensor(xs, device=dev), torch.tensor(zs, device=dev), axes_t)
            c0, g0 = rec.evaluate(r_np)
            want_c = osph.nested_sphere_reconstruction_cost(xs, zs, axes_np, r_np)
            note("nested_sphere_recon_cost", abs(c0 - want_c) / max(1.0, abs(want_c)))
            kk = int(rng.integers(0, L))
            hh = 1e-6
            rp, rm = r_np.copy(), r_np.copy()
            rp[kk] += hh
            rm[kk] -= hh
            fdv = (osph.nested_sphere_reconstruction_cost(xs, zs, axes_np, rp) - osph.nested_sphere_reconstruction_cost(xs, zs, axes_np, rm)) / (2 * hh)
            note("nested_sphere_recon_fd", abs(g0[kk] - fdv) / max(1.0, abs(fdv)))
    for k in sorted(worst):
        print(f"{k:28s} worst {worst[k]:.2e}")
    bad = {k: v for k, v in worst.items() if v > (1e-5 if "fd" in k else (1e-7 if "gp_mll" in k else 1e-9))}
    print("FAIL" if bad else "OK", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
