"""Development probe: affine-invariant distances between ill-conditioned / graded SPD matrices (every dispatch of gabo_spd_ai_pairwise) against the oracle.
The eigenvalue iterations of the pairwise kernels (spd_eig.hpp root-free QL per lane, wave_eigh above the register limits) deflate from one end of the
tridiagonal form; graded congruence matrices are the inputs on which such an iteration can run out of sweeps."""
import numpy as np
import torch
from gabotorch_amd import _lib, ops
from oracle import spd as ospd

rng = np.random.default_rng(11)
for d in (2, 3, 4, 5, 8, 10, 12, 13, 16, 17, 20, 24):
    n = 48
    worst = 0.0
    for kind in ("wishart", "decades", "graded_down", "graded_up", "cluster_small"):
        def make(m):
            q = np.linalg.qr(rng.standard_normal((m, d, d)))[0]
            if kind == "wishart":
                g = rng.standard_normal((m, d, d)); return np.einsum("nab,ncb->nac", g, g) / d + 0.05 * np.eye(d)
            if kind == "decades":
                lam = 10.0 ** rng.uniform(-5, 5, (m, d)); return np.einsum("nab,nb,ncb->nac", q, lam, q)
            if kind == "cluster_small":
                lam = np.concatenate([np.full((m, d - 1), 1e-6) * (1 + 1e-3 * rng.uniform(size=(m, d - 1))), np.ones((m, 1))], axis=1)
                return np.einsum("nab,nb,ncb->nac", q, lam, q)
            g = rng.standard_normal((m, d, d)); w = np.einsum("nab,ncb->nac", g, g) / d + 0.05 * np.eye(d)
            gr = 10.0 ** (-np.arange(d) * rng.uniform(0.1, 0.6, (m, 1)))
            if kind == "graded_up": gr = gr[:, ::-1]
            return np.einsum("na,nab,nb->nab", gr, w, gr)
        a, b = make(n), make(n)
        a = 0.5 * (a + a.transpose(0, 2, 1)); b = 0.5 * (b + b.transpose(0, 2, 1))
        want = ospd.affine_invariant_distance(a, b)
        va = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(a), device="cuda:0")
        vb = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(b), device="cuda:0")
        got = ops.spd_ai_pairwise(va, vb, 1.0, _lib.GABO_OUT_DISTANCE).cpu().numpy()
        # the distance of a pair with condition number c is known to ~eps c / dist (logarithms of eigenvalues known to eps |M|)
        lam = np.linalg.eigvalsh(ospd.congruence_matrices(a, b))
        cond = lam.max(-1) / lam.min(-1)
        err = np.abs(got - want) / want
        tol = 1e-10 + 4e-16 * cond
        ratio = (err / tol).max()
        worst = max(worst, ratio)
        flag = "  <-- FAIL" if ratio > 1 else ""
        print(f"d {d:2d} {kind:14s} max rel err {err.max():.2e}  (cond up to {cond.max():.1e}; err / tol {ratio:.2e}){flag}")
