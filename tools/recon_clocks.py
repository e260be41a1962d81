#!/usr/bin/env python3
"""Shader-cycle anatomy of one fused reconstruction evaluation (block 0 of gabo_nested_spd_reconstruction) at D = 20 -> 2.
Needs the instrumented build:  python tools/ab_build.py clk nested_spd_reconstruction.hip -DGABO_RECON_CLOCKS
then  GABO_HIP_LIB=gabotorch_amd/libgabo_hip_clk.so python tools/recon_clocks.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _lib, ops                                                     # noqa: E402


def main():
    rng = np.random.default_rng(3)
    D, d, N = 20, 2, 10
    m = D - d
    q = np.linalg.qr(rng.standard_normal((N, D, D)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.1, 5.0, (N, D)), q)
    X = 0.5 * (X + X.transpose(0, 2, 1))
    R = np.linalg.qr(rng.standard_normal((D, D)))[0]
    W, V = R[:, :d], R[:, d:]
    Y = np.einsum("ab,nac,cd->nbd", W, X, W)
    qc = np.linalg.qr(rng.standard_normal((m, m)))[0]
    C = (qc * rng.uniform(0.5, 2.0, m)) @ qc.T
    K = rng.standard_normal((d, m))
    K *= 0.4 / np.linalg.norm(K)
    T = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda:0")   # noqa: E731
    lib = _lib.load()
    names = ["load", "eig C", "C^1/2, B, Xrec", "eig Xrec", "logm, adjoint", "partials", "last block"]
    for metric, tag in ((_lib.GABO_RECON_LOG_EUCLIDEAN, "log-Euclidean"), (_lib.GABO_RECON_AFFINE_INVARIANT, "affine-invariant")):
        rec = ops.NestedSpdReconstruction(T(X), T(Y), T(W), metric)
        for grad in (True, False):
            for _ in range(3):
                rec.evaluate_host(V, C, K, grad=grad)
            buf = (ctypes.c_longlong * 16)()
            lib.gabo_debug_recon_clocks.argtypes = [ctypes.c_void_p]
            lib.gabo_debug_recon_clocks(buf)
            t = list(buf)
            parts = ", ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names) if grad or i < 6)
            print(f"{tag}, {'value + gradient' if grad else 'value only'}: {parts}; block 0 total {t[7 if grad else 6] - t[0]} cycles")


if __name__ == "__main__":
    main()
