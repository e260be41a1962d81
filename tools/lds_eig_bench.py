#!/usr/bin/env python3
"""Wave-per-matrix eigen-solver (lds_jacobi up to d = 8, wave_eigh above) timings: logm of N SPD matrices by dimension.   python tools/lds_eig_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _lib, ops                                                     # noqa: E402


def main():
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    for d in (2, 5, 8, 9, 10, 12, 13, 16, 17, 18, 20, 21, 24, 28, 29, 32):
        for n in (32, 4096):
            q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
            m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.1, 5.0, (n, d)), q)
            x = torch.tensor(0.5 * (m + m.transpose(0, 2, 1)), device=dev)
            ref = np.linalg.eigh(m)
            want = np.einsum("nab,nb,ncb->nac", ref[1], np.log(ref[0]), ref[1])
            got = ops.spd_matrix_function(x, _lib.GABO_SPD_LOGM)
            err = float(np.abs(got.cpu().numpy() - want).max())
            for _ in range(3):
                ops.spd_matrix_function(x, _lib.GABO_SPD_LOGM)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                ops.spd_matrix_function(x, _lib.GABO_SPD_LOGM)
            b.record()
            torch.cuda.synchronize()
            print(f"d={d:3d} n={n:5d}  logm {a.elapsed_time(b) / 20 * 1e3:9.1f} us/launch   max err {err:.1e}")


if __name__ == "__main__":
    main()
