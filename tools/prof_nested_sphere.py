#!/usr/bin/env python3
"""Driver for rocprofv3 on the nested-sphere chain (D = 51 -> 3, 48 levels): a few launches of the projection of 4096 points and of the
reconstruction evaluation (64 data points, values + gradients).   rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- python tools/prof_nested_sphere.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gabotorch_amd import _lib, ops                      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
D, lat, n = 51, 3, 4096
L = D - lat
rng = np.random.default_rng(5151)
axes = []
for k in range(L):
    a = rng.standard_normal(D - k)
    axes.append(torch.tensor(a / np.linalg.norm(a), device=dev))
r = rng.uniform(0.8, 2.2, L)
x = rng.standard_normal((n, D))
x /= np.linalg.norm(x, axis=1, keepdims=True)
xt = torch.tensor(x, device=dev)
frames, dists = ops._nested_sphere_frames(axes, list(r), D, dev)
z = torch.empty(n, lat, dtype=torch.float64, device=dev)
for _ in range(5):
    lib.gabo_nested_sphere_project(xt.data_ptr(), frames.data_ptr(), dists.data_ptr(), z.data_ptr(), None, n, D, L, ops._stream_ptr(dev))
torch.cuda.synchronize()
rec = ops.NestedSphereReconstruction(xt[:64], z[:64], axes)
for _ in range(5):
    rec.evaluate(r)
torch.cuda.synchronize()
