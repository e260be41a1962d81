"""Oracle (CPU, numpy/scipy) for the step that consumes the kernel strip in the acquisition maximiser: exact-GP posterior and
analytic acquisition values.  Test infrastructure only - see oracle/__init__.py.

[3P] botorch / gpytorch semantics (unpinned: the packages are absent from the image):
  eval-mode ExactGP:  mean = m + k*^T (K + s2 I)^-1 (y - m),  var = k** - k*^T (K + s2 I)^-1 k*,  K = outputscale * base kernel
  botorch.acquisition.ExpectedImprovement(model, best_f, maximize):  sigma = sqrt(clamp_min(var, 1e-9)),
      u = (mean - best_f) / sigma (negated when maximize=False),  EI = sigma (phi(u) + u Phi(u))
  call sites in the reference: examples/gabo_spd.py:197 (EI, maximize=False), manifold_optimize.py:182-184 (cost = -acq)."""
import numpy as np
from scipy.stats import norm


def gp_posterior(k_train, k_star, k_star_star, y, mean, outputscale, noise):
    """k_train (n, n), k_star (r, n), k_star_star (r,): BASE kernel values.  -> posterior mean (r,), variance (r,)."""
    kt = outputscale * np.asarray(k_train, dtype=np.float64) + noise * np.eye(len(y))
    ks = outputscale * np.asarray(k_star, dtype=np.float64)
    sol = np.linalg.solve(kt, ks.T)                       # (n, r)
    mu = mean + ks @ np.linalg.solve(kt, np.asarray(y, dtype=np.float64) - mean)
    var = outputscale * np.asarray(k_star_star, dtype=np.float64) - np.sum(ks * sol.T, axis=1)
    return mu, var


def expected_improvement(mu, var, best_f, maximize):
    sigma = np.sqrt(np.maximum(var, 1e-9))
    u = (mu - best_f) / sigma
    if not maximize:
        u = -u
    return sigma * (norm.pdf(u) + u * norm.cdf(u))


def marginal_log_likelihood(e, y, theta, outputscale, noise, mean):
    """Exact-GP log marginal likelihood log N(y | mean, Ky), Ky = outputscale * exp(-theta * e) + noise * I, and its gradient with
    respect to (theta, outputscale, noise, mean): [3P] the quantity gpytorch's ExactMarginalLogLikelihood differentiates (before its
    priors and its division by n) inside fit_gpytorch_model (call site: examples/bo_spd/benchmark_examples/gabo_spd.py:194).
    Textbook identities (Rasmussen & Williams eq. 5.8-5.9): d ll / d p = tr((alpha alpha^T - Ky^-1) dKy/dp) / 2.
    -> (ll, grad (4,))."""
    e = np.asarray(e, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    kb = np.exp(-theta * e)
    ky = outputscale * kb + noise * np.eye(n)
    chol = np.linalg.cholesky(ky)
    r = y - mean
    alpha = np.linalg.solve(ky, r)
    ll = -0.5 * r @ alpha - np.log(np.diag(chol)).sum() - 0.5 * n * np.log(2.0 * np.pi)
    w = np.outer(alpha, alpha) - np.linalg.inv(ky)
    grad = np.array([0.5 * np.sum(w * (-outputscale * e * kb)), 0.5 * np.sum(w * kb), 0.5 * np.trace(w), alpha.sum()])
    return ll, grad
