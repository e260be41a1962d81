"""The reference ITSELF timed on the config-3 / config-2 inputs of bench.py (build container only - /root/reference does not
exist on the GPU box and nothing here is imported by the product, the tests or bench.py).

    python -W ignore tools/time_reference_cpu.py [--rows 4096] [--tile 256] [--out profiles/r04_reference_cpu.json]

SPD (config 3): vector_to_symmetric_matrix_mandel_torch (Riemannian_utils/spd_utils_torch.py:159-194, its per-vector Python loop) on x1 and x2,
affine_invariant_distance_torch (:53-121, its per-pair symeig loop) and exp(-beta d^2) (kernel_utils/kernels_spd.py:91-100), called on row tiles of
x1 (`--tile` rows x all N columns per call: the un-tiled call materialises 4 N^2 d^2 doubles, >= 54 GB at N = 4096), all torch threads.
Sphere (config 2): sphere_distance_torch (Riemannian_utils/sphere_utils_torch.py:12-55) + exp, un-tiled.
Only shim: torch.symeig (removed from torch) -> torch.linalg.eigh(UPLO='U'), SURVEY App. D.
`--rows R` times the first R rows of x1 against all N columns (R = N: the whole Gram); the rate is pairs / wall-clock."""
import argparse
import collections
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = collections.namedtuple("symeig", ["eigenvalues", "eigenvectors"])
torch.symeig = lambda A, eigenvectors=False, upper=True: R(*torch.linalg.eigh(A, UPLO="U" if upper else "L"))
sys.path.insert(0, "/root/reference")
from BoManifolds.Riemannian_utils.spd_utils_torch import affine_invariant_distance_torch, vector_to_symmetric_matrix_mandel_torch  # noqa: E402
from BoManifolds.Riemannian_utils.sphere_utils_torch import sphere_distance_torch  # noqa: E402

import bench  # noqa: E402  (the synthetic inputs of the bench line: same generator, same seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=bench.N_POINTS)
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-sphere", action="store_true")
    args = ap.parse_args()
    n, d = bench.N_POINTS, bench.DIM
    x = torch.tensor(bench.synthetic_spd_mandel(n, d, 1234))
    beta = bench.BETA
    rows = min(args.rows, n)
    res = {"host": {"cores": os.cpu_count(), "torch_threads": torch.get_num_threads()}, "torch": torch.__version__}
    t0 = time.perf_counter()
    x2m = vector_to_symmetric_matrix_mandel_torch(x)                     # the reference converts both arguments on every forward call
    t_mandel2 = time.perf_counter() - t0
    t_mandel1 = t_dist = t_exp = 0.0
    check = None
    for lo in range(0, rows, args.tile):
        hi = min(lo + args.tile, rows)
        t0 = time.perf_counter()
        x1m = vector_to_symmetric_matrix_mandel_torch(x[lo:hi])
        t1 = time.perf_counter()
        dist = affine_invariant_distance_torch(x1m, x2m)
        t2 = time.perf_counter()
        k = torch.exp(-beta * dist.double() ** 2)
        t3 = time.perf_counter()
        t_mandel1 += t1 - t0
        t_dist += t2 - t1
        t_exp += t3 - t2
        if check is None:
            check = k[:8, :8].numpy().copy()
        print(f"rows {lo}-{hi}: distance {t2 - t1:.1f} s ({(hi - lo) * n / (t2 - t1):.3e} pairs/s)", flush=True)
    total = t_mandel2 + t_mandel1 + t_dist + t_exp
    pairs = rows * n
    res["spd_config3"] = {"workload": f"reference forward on rows 0..{rows} x all {n} columns of the bench input (N={n}, d={d}, seed 1234), x1 tiles of {args.tile} rows",
                          "pairs": pairs, "seconds": total, "pairs_per_s": pairs / total,
                          "seconds_mandel_x2": t_mandel2, "seconds_mandel_x1": t_mandel1, "seconds_distance": t_dist, "seconds_exp": t_exp,
                          "k_block_8x8": check.tolist()}
    from oracle import spd as ospd
    want = ospd.spd_ai_gaussian_kernel(x[:8].numpy(), x[:8].numpy(), beta)
    res["spd_config3"]["max_rel_diff_oracle_vs_reference_8x8"] = float(np.max(np.abs(check - want) / np.abs(want)))
    if not args.no_sphere:
        rng = np.random.default_rng(1234)
        s = rng.standard_normal((n, 10))
        s /= np.linalg.norm(s, axis=1, keepdims=True)
        st = torch.tensor(s)
        sbeta = 0.6 + float(np.log(2.0))
        t0 = time.perf_counter()
        ks = torch.exp(-sbeta * sphere_distance_torch(st, st).double() ** 2)
        ts = time.perf_counter() - t0
        res["sphere_config2"] = {"workload": f"reference sphere_distance_torch + exp, N={n}, ambient dimension 10, un-tiled", "pairs": n * n, "seconds": ts,
                                 "pairs_per_s": n * n / ts, "k00": float(ks[0, 0])}
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
