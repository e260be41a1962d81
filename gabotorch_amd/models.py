"""Minimal exact-GP surrogate + analytic acquisition with botorch's call conventions, so the acquisition sweep runs where
gpytorch/botorch are absent (this image, the GPU box).  When those packages exist, their `SingleTaskGP` /
`ExpectedImprovement` can be used with the kernel classes of this package instead; `joint_optimize_manifold` only needs
a callable `acq(X: b x q x d) -> b`.

Semantics restated from botorch/gpytorch ([3P] unpinned): constant mean, ScaleKernel outputscale,
homoskedastic Gaussian noise; eval-mode posterior evaluates k(X*, X_train) with the CANDIDATES as the first argument;
EI(maximize=False): sigma = sqrt(clamp_min(var, 1e-9)), u = -(mu - best_f)/sigma, EI = sigma (phi(u) + u Phi(u)).
The small dense linear algebra here (n_train <= a few hundred) is torch; the kernel evaluations and their gradients are
the HIP kernels.
"""
import math
import warnings

import torch


class BadInitialCandidatesWarning(RuntimeWarning):
    pass


class ExactGP(torch.nn.Module):
    def __init__(self, train_x, train_y, base_kernel, outputscale=1.0, noise=1e-2, mean=None):
        super().__init__()
        self.base_kernel = base_kernel
        self.train_x = train_x.double()
        y = train_y.double().reshape(-1)
        self.train_y = y
        self.outputscale = float(outputscale)
        self.noise = float(noise)
        self.mean = float(y.mean()) if mean is None else float(mean)
        self._cache = None

    def _train_cache(self):
        if self._cache is None:
            with torch.no_grad():
                k = self.outputscale * self.base_kernel.forward(self.train_x, self.train_x)
                n = k.shape[-1]
                k = k + self.noise * torch.eye(n, dtype=k.dtype, device=k.device)
                L = torch.linalg.cholesky(k)
                alpha = torch.cholesky_solve((self.train_y.to(k.device) - self.mean).unsqueeze(-1), L).squeeze(-1)
            self._cache = (L, alpha)
        return self._cache

    def posterior(self, X):
        """X: b x 1 x d  ->  (mean b, variance b)."""
        if X.dim() == 2:
            X = X.unsqueeze(-2)
        b = X.shape[0]
        L, alpha = self._train_cache()
        xt = self.train_x.to(X.device).expand(b, *self.train_x.shape)           # stride-0 batch: factored once by the kernel
        ks = self.outputscale * self.base_kernel.forward(X, xt)                 # b x 1 x n   (candidates first)
        kss = self.outputscale * self.base_kernel.forward(X, X)                 # b x 1 x 1
        ks = ks.squeeze(-2)
        Ld, ad = L.to(ks.device), alpha.to(ks.device)
        mean = self.mean + ks @ ad
        v = torch.linalg.solve_triangular(Ld, ks.transpose(-1, -2), upper=False)  # n x b
        var = kss.reshape(b) - (v * v).sum(0)
        return mean, var


class ExpectedImprovement(torch.nn.Module):
    """botorch.acquisition.ExpectedImprovement(model, best_f, maximize) [3P]."""

    is_nonnegative = True

    def __init__(self, model, best_f, maximize=True):
        super().__init__()
        self.model = model
        self.best_f = float(best_f)
        self.maximize = maximize

    def forward(self, X):
        mean, var = self.model.posterior(X)
        sigma = var.clamp_min(1e-9).sqrt()
        u = (mean - self.best_f) / sigma
        if not self.maximize:
            u = -u
        pdf = torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        return sigma * (pdf + u * cdf)


class PosteriorMean(torch.nn.Module):
    def __init__(self, model, maximize=True):
        super().__init__()
        self.model, self.maximize = model, maximize

    def forward(self, X):
        mean, _ = self.model.posterior(X)
        return mean if self.maximize else -mean


def is_nonnegative(acq_function):
    return bool(getattr(acq_function, "is_nonnegative", False))


def initialize_q_batch(X, Y, n, eta=1.0):
    """botorch.optim.initializers.initialize_q_batch [3P]: Boltzmann sampling on standardised values, argmax forced in."""
    n_samples = X.shape[0]
    if n > n_samples:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({n_samples})")
    if n == n_samples:
        return X
    Ystd = Y.std()
    if Ystd == 0:
        warnings.warn("All acquisition values for raw samples points are the same.", BadInitialCandidatesWarning)
        return X[torch.randperm(n=n_samples, device=X.device)][:n]
    max_val, max_idx = torch.max(Y, dim=0)
    Z = (Y - Y.mean()) / Ystd
    etaZ = eta * Z
    weights = torch.exp(etaZ)
    while torch.isinf(weights).any():
        etaZ *= 0.5
        weights = torch.exp(etaZ)
    idcs = torch.multinomial(weights, n)
    if max_idx not in idcs:
        idcs[-1] = max_idx
    return X[idcs]


def initialize_q_batch_nonneg(X, Y, n, eta=1.0, alpha=1e-4):
    """botorch.optim.initializers.initialize_q_batch_nonneg [3P]."""
    n_samples = X.shape[0]
    if n > n_samples:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({n_samples})")
    if n == n_samples:
        return X
    max_val, max_idx = torch.max(Y, dim=0)
    if torch.any(max_val <= 0):
        warnings.warn("All acquisition values for raw sampled points are nonpositive, so initial conditions are being "
                      "selected randomly.", BadInitialCandidatesWarning)
        return X[torch.randperm(n=n_samples, device=X.device)][:n]
    pos = Y > 0
    num_pos = int(pos.sum().item())
    if num_pos < n:
        remaining = n - num_pos
        rand_idx = torch.randperm(n_samples - num_pos, device=Y.device)[:remaining]
        xpos, xneg = X[pos], X[~pos][rand_idx]
        return torch.cat([xpos, xneg], dim=0)[torch.randperm(n, device=X.device)]
    alpha_pos = Y >= alpha * max_val
    while alpha_pos.sum() < n:
        alpha = 0.1 * alpha
        alpha_pos = Y >= alpha * max_val
    alpha_pos_idcs = torch.arange(len(Y), device=Y.device)[alpha_pos]
    weights = torch.exp(eta * (Y[alpha_pos] / max_val - 1))
    idcs = alpha_pos_idcs[torch.multinomial(weights, n)]
    if max_idx not in idcs:
        idcs[-1] = max_idx
    return X[idcs]


def get_best_candidates(batch_candidates, batch_values):
    """botorch.generation.gen.get_best_candidates [3P]: argmax over restarts (ties -> lowest index)."""
    best = torch.argmax(batch_values.view(-1), dim=0)
    return batch_candidates[best]
