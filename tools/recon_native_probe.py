#!/usr/bin/env python3
"""Where one optimisation of the reconstruction parameters spends its time with the native host loop (gabo_nested_spd_reconstruction_solve):
set-up (prepare launch, staging buffers, the start candidates), the loop itself (launches x launch time + host C++), by ambient dimension.
   python tools/recon_native_probe.py [D ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gabotorch_amd import ops                                                                     # noqa: E402
from gabotorch_amd.manifold_optimization.conjugate_gradient import ConjugateGradient             # noqa: E402
from gabotorch_amd.nested_mappings import nested_spd_optimization as nso                          # noqa: E402
from gabotorch_amd.nested_mappings.nested_spd_utils import projection_from_spd_to_nested_spd      # noqa: E402


def main():
    dev = "cuda:0"
    ops.set_error_checking(False)
    for D in [int(a) for a in sys.argv[1:]] or [5, 10, 20]:
        rng = np.random.default_rng(D)
        N, d = 10, 2
        A = rng.standard_normal((N, D, D))
        X = torch.tensor(A @ A.transpose(0, 2, 1) / D + np.eye(D), device=dev)
        W = torch.tensor(np.linalg.qr(rng.standard_normal((D, d)))[0], device=dev)
        Y = projection_from_spd_to_nested_spd(X, W)
        rows = []
        for native in (True, False, True, False, True):
            np.random.seed(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nso.optimize_reconstruction_parameters_nested_spd(X, Y, W, ConjugateGradient(maxiter=100), nb_init_candidates=20, maxiter=6,
                                                              cost_function=nso.min_log_euclidean_distance_reconstruction_cost, native=native)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            log = nso.optimize_reconstruction_parameters_nested_spd.last_log
            rows.append((native, wall, log))
        for native, wall, log in rows[1:]:
            extra = f"launches {log['launches']} evaluations {log['evaluations']} inner iterations {log['inner_iterations']} loop {1e3 * log['time']:.1f} ms" if native \
                else f"loop {1e3 * log['time']:.1f} ms"
            print(f"D={D:2d} native={native!s:5}: wall {1e3 * wall:.1f} ms, outer {log['iterations']} ({log['stop_reason']}), final cost {log['final_cost']:.6f}, {extra}")


if __name__ == "__main__":
    main()
