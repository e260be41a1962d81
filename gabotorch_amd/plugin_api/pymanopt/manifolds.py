"""pymanopt.manifolds: the HIP-backed manifolds of this package (always - they are what the lock-step solvers batch over), plus the host
manifolds the surrogate fit on product manifolds uses."""
from ...manifold_optimization.host_manifolds import Euclidean, Grassmann, Product  # noqa: F401
from ...manifolds import PositiveDefinite, Sphere  # noqa: F401
