"""pymanopt.solvers: stock `TrustRegions` (examples/gabo_sphere.py:151) = the robust variant of this package; `ConjugateGradient`
(manifold_gp_fit.py) = the restated Riemannian CG."""
from ...manifold_optimization.conjugate_gradient import ConjugateGradient  # noqa: F401
from ...manifold_optimization.robust_trust_regions import TrustRegions  # noqa: F401
