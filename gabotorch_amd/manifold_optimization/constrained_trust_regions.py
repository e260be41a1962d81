"""`ConstrainedTrustRegions` and `StrictConstrainedTrustRegions` under the reference's module path and constructors
(BoManifolds/manifold_optimization/constrained_trust_regions.py:75-117, 737-780).

Both are `BatchedTrustRegions`: equality / inequality constraints are linearised inside the truncated CG, which stops where the
violation reaches `Delta_cons` (default 1e-6, :139-141); the strict variant additionally rejects every proposal that violates a
constraint at the proposed point itself (:932-951).  `solve(problem, x=None, eq_constraints=None, ineq_constraints=None, mininner=1,
maxinner=None, Delta_bar=None, Delta0=None, Delta_cons=None)` as in the reference; single constraints or lists are accepted (:151-159)."""
from .batched_trust_regions import BatchedTrustRegions


class ConstrainedTrustRegions(BatchedTrustRegions):
    def __init__(self, miniter=3, kappa=0.1, theta=1.0, rho_prime=0.1, use_rand=False, rho_regularization=1e3, *args, **kwargs):
        kwargs.pop("strict_constraints", None)
        super().__init__(miniter, kappa, theta, rho_prime, use_rand, rho_regularization, *args, strict_constraints=False, **kwargs)


class StrictConstrainedTrustRegions(BatchedTrustRegions):
    def __init__(self, miniter=3, kappa=0.1, theta=1.0, rho_prime=0.1, use_rand=False, rho_regularization=1e3, *args, **kwargs):
        kwargs.pop("strict_constraints", None)
        super().__init__(miniter, kappa, theta, rho_prime, use_rand, rho_regularization, *args, strict_constraints=True, **kwargs)
