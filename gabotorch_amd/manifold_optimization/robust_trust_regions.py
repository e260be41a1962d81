"""`TrustRegions` under the reference's module path and constructor (BoManifolds/manifold_optimization/robust_trust_regions.py:71-110:
pymanopt's Riemannian trust regions + truncated CG with the guard against d_Hd == 0).

The state machine itself is `BatchedTrustRegions` (all restarts in lock step, device-resident where the problem allows); this class
only fixes the reference's name, keyword arguments and `solve(problem, x=None, mininner=1, maxinner=None, Delta_bar=None,
Delta0=None)` signature.  `solve` takes what the reference's solver takes - a pymanopt-style problem and ONE starting point as a numpy
array (manifold_optimize.py:217) - as well as the library's batched problems with an R x ... tensor of starting points."""
from .batched_trust_regions import BatchedTrustRegions


class TrustRegions(BatchedTrustRegions):
    def __init__(self, miniter=3, kappa=0.1, theta=1.0, rho_prime=0.1, use_rand=False, rho_regularization=1e3, *args, **kwargs):
        kwargs.pop("strict_constraints", None)
        super().__init__(miniter, kappa, theta, rho_prime, use_rand, rho_regularization, *args, strict_constraints=False, **kwargs)

    def solve(self, problem, x=None, mininner=1, maxinner=None, Delta_bar=None, Delta0=None):
        return super().solve(problem, x, None, None, mininner, maxinner, Delta_bar, Delta0, None)
