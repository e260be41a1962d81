#!/usr/bin/env python3
"""Host-side profile (cProfile) of the PYTHON loop of optimize_reconstruction_parameters_nested_spd (native=False) at D = 20 -> 2, N = 10:
where an augmented-Lagrangian run spends its wall-clock between the fused launch, the pinned copies and the numpy manifold arithmetic - the
measurement that motivated the native loop (tools/recon_native_probe.py times that one).   python tools/recon_profile.py"""
import os, sys, cProfile, pstats, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from gabotorch_amd.nested_mappings import nested_spd_optimization as nso
from gabotorch_amd.manifold_optimization.conjugate_gradient import ConjugateGradient
rng = np.random.default_rng(3)
D, d, N = int(os.environ.get("RECON_D", "20")), 2, 10
q = np.linalg.qr(rng.standard_normal((N, D, D)))[0]
X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.1, 5.0, (N, D)), q); X = 0.5*(X+X.transpose(0,2,1))
W = np.linalg.qr(rng.standard_normal((D, D)))[0][:, :d]
Y = np.einsum("ab,nac,cd->nbd", W, X, W)
T = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda:0")
np.random.seed(0)
def go():
    return nso.optimize_reconstruction_parameters_nested_spd(T(X), T(Y), T(W), ConjugateGradient(maxiter=100), cost_function=nso.min_log_euclidean_distance_reconstruction_cost, nb_init_candidates=20, maxiter=6, native=False)
go()
t=time.perf_counter(); go(); print("wall", time.perf_counter()-t)
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
