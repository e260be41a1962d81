"""Module path of the reference's nested SPD kernels (`BoManifolds/kernel_utils/kernels_nested_spd.py:19-246`): the examples import
`NestedSpdLogEuclideanGaussianKernel` / `NestedSpdAffineInvariantGaussianKernel` from here (`examples/hd_bo_spd/benchmark_examples/hd_gabo_spd.py`).
The classes are defined next to their base kernels in `kernels_spd.py` (projection Y = W^T X W by `gabo_spd_project`, then the base Gram kernel)."""
from .kernels_spd import NestedSpdAffineInvariantGaussianKernel, NestedSpdLogEuclideanGaussianKernel

__all__ = ["NestedSpdAffineInvariantGaussianKernel", "NestedSpdLogEuclideanGaussianKernel"]
