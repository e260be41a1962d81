#!/usr/bin/env python3
"""Wall-clock phases of one HD-GaBO iteration (examples/hd_gabo_spd.py: surrogate fit on the Grassmannian, reconstruction parameters by the
augmented Lagrangian, latent acquisition sweep with the original-space eigenvalue constraints, objective) by ambient dimension.
   python tools/hd_gabo_breakdown.py [--dims 5 10 20] [--iters 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[5, 10, 20])
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    import hd_gabo_spd
    for D in a.dims:
        t = {}
        hd_gabo_spd.run(dim=D, latent=2, iters=a.iters, verbose=False, timings=t)
        # the first iteration pays one-off costs (library load, allocator, graph captures): medians over the rest
        med = {k: 1e3 * float(np.median(v[1:])) for k, v in t.items()}
        total = sum(med.values())
        print(f"HD-GaBO D={D:2d} -> 2, n = 6..{5 + a.iters - 1}: " + ", ".join(f"{k} {v:.1f} ms" for k, v in med.items()) + f"; iteration {total:.1f} ms")


if __name__ == "__main__":
    main()
