#!/usr/bin/env python3
"""Launch time of gabo_gp_mll by training-set size, and the surrogate fit with and without it.   python tools/gp_mll_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _lib, models, ops                                                     # noqa: E402
from gabotorch_amd._compat import ScaleKernel                                                   # noqa: E402
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel             # noqa: E402
from gabotorch_amd.Riemannian_utils.spd_utils import spd_sample, symmetric_matrix_to_vector_mandel    # noqa: E402


class _Man:
    _n, min_eig, max_eig = 5, 0.05, 5.0


def main():
    dev = "cuda:0"
    np.random.seed(0)
    lib = _lib.load()
    for n in (8, 16, 35, 55, 64, 96, 128, 160, 200, 256, 512, 1024):
        x = torch.tensor(np.stack([symmetric_matrix_to_vector_mandel(spd_sample(_Man())) for _ in range(n)]), device=dev)
        y = torch.randn(n, dtype=torch.float64, device=dev)
        d = ops.spd_ai_pairwise(x, x, 1.0, _lib.GABO_OUT_DISTANCE)
        e = (d * d).contiguous()
        out = torch.empty(6, dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        if n <= _lib.GABO_GP_MLL_MAX_N:
            call = lambda: lib.gabo_gp_mll(e.data_ptr(), y.data_ptr(), n, 0.9, 1.0, 0.02, 0.0, out.data_ptr(), st)    # noqa: E731
        else:                                  # the tiled sweep (gabo_gp_mll_large): 2 launches per 32 pivots
            wsb = int(lib.gabo_gp_mll_large_workspace_bytes(n))
            ws = torch.empty(wsb // 8 + 1, dtype=torch.float64, device=dev)
            call = lambda: lib.gabo_gp_mll_large(e.data_ptr(), y.data_ptr(), n, 0.9, 1.0, 0.02, 0.0, 0, out.data_ptr(), None, ws.data_ptr(), wsb, st)    # noqa: E731
        for _ in range(5):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            call()
        b.record()
        torch.cuda.synchronize()
        line = f"n={n:4d}  gabo_gp_mll{'' if n <= _lib.GABO_GP_MLL_MAX_N else '_large'} {a.elapsed_time(b) / 200 * 1e3:7.1f} us/evaluation"
        for fast in (True, False):
            ts = []
            for _ in range(4 if n <= 160 else 2):
                gp = models.SingleTaskGP(x, y, ScaleKernel(SpdAffineInvariantGaussianKernel(beta_min=0.25),
                                                           outputscale_prior=models.GammaPrior(2.0, 0.15)),
                                         noise_prior=models.GammaPrior(1.1, 0.05))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                models.fit_gpytorch_model(gp, fast=fast)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            line += f"   fit({'one launch per evaluation' if fast else 'autograd'}) {min(ts) * 1e3:7.2f} ms"
        print(line)


if __name__ == "__main__":
    main()
