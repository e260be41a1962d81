"""Module path of the reference's solver (`BoManifolds/manifold_optimization/augmented_Lagrange_method.py:29`, capital L), imported by
`examples/bo_sphere/constrained_benchmark_examples/gabo_sphere_{equality,inequality}_constraints.py`; implementation in
`augmented_lagrange_method.py`."""
from .augmented_lagrange_method import AugmentedLagrangeMethod

__all__ = ["AugmentedLagrangeMethod"]
