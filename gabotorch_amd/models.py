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
        for p in (self.base_kernel.parameters() if hasattr(self.base_kernel, "parameters") else ()):     # fixed hyper-parameters: no gradient (and no device->host copy of one) in the acquisition path
            p.requires_grad_(False)

    def _train_cache(self):
        if self._cache is None:
            with torch.no_grad():
                k = self.outputscale * self.base_kernel.forward(self.train_x, self.train_x)
                n = k.shape[-1]
                k = k + self.noise * torch.eye(n, dtype=k.dtype, device=k.device)
                L = torch.linalg.cholesky(k)
                alpha = torch.cholesky_solve((self.train_y.to(k.device) - self.mean).unsqueeze(-1), L).squeeze(-1)
                # L^-1 once: the per-call triangular solve becomes a GEMM (rocBLAS trsm allocates workspace, which a hipGraph
                # capture of the acquisition evaluation does not allow)
                Linv = torch.linalg.solve_triangular(L, torch.eye(n, dtype=L.dtype, device=L.device), upper=False)
            self._cache = (Linv, alpha)
        return self._cache

    def posterior(self, X):
        """X: b x 1 x d  ->  (mean b, variance b)."""
        if X.dim() == 2:
            X = X.unsqueeze(-2)
        b = X.shape[0]
        Linv, alpha = self._train_cache()
        xt = self.train_x.to(X.device).expand(b, *self.train_x.shape)           # stride-0 batch: factored once by the kernel
        ks = self.outputscale * self.base_kernel.forward(X, xt)                 # b x 1 x n   (candidates first)
        kss = self.outputscale * self.base_kernel.forward(X, X)                 # b x 1 x 1
        ks = ks.squeeze(-2)
        Li, ad = Linv.to(ks.device), alpha.to(ks.device)
        mean = self.mean + ks @ ad
        v = ks @ Li.transpose(-1, -2)                                           # b x n  = (L^-1 ks^T)^T
        var = kss.reshape(b) - (v * v).sum(-1)
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


# ---------------------------------------------------------------------------------------------- trainable surrogate
class GammaPrior:
    """Gamma(concentration, rate) log-density (gpytorch.priors.GammaPrior semantics) [3P]."""

    def __init__(self, concentration, rate):
        self.concentration, self.rate = float(concentration), float(rate)

    def log_prob(self, x):
        a, b = self.concentration, self.rate
        return a * math.log(b) + (a - 1.0) * torch.log(x) - b * x - math.lgamma(a)

    @property
    def mode(self):
        return max((self.concentration - 1.0) / self.rate, 0.0)


class SingleTaskGP(torch.nn.Module):
    """Constant-mean exact GP with a Gaussian likelihood and trainable hyper-parameters, laid out like the models of the
    reference examples (examples/gabo_spd.py:165-176: ScaleKernel(base, outputscale_prior=Gamma(2, .15)), noise prior
    Gamma(1.1, .05), noise constraint GreaterThan(1e-8), initial noise = prior mode).  Priors registered on the kernels through
    the gpytorch-compatible `register_prior` are picked up when the stand-in Kernel base class is in use."""

    def __init__(self, train_x, train_y, covar_module, noise_prior=None, noise_lower_bound=1e-8):
        super().__init__()
        self.train_x = train_x.double()
        self.train_y = train_y.double().reshape(-1)
        self.covar_module = covar_module
        self.noise_prior = noise_prior
        self.noise_lower_bound = float(noise_lower_bound)
        init_noise = noise_prior.mode if noise_prior is not None and noise_prior.mode > noise_lower_bound else 1e-2
        self.raw_noise = torch.nn.Parameter(torch.tensor(math.log(math.expm1(init_noise - self.noise_lower_bound)), dtype=torch.float64))
        self.mean_constant = torch.nn.Parameter(torch.zeros((), dtype=torch.float64))
        self._cache = None

    @property
    def noise(self):
        return torch.nn.functional.softplus(self.raw_noise) + self.noise_lower_bound

    def _kxx(self):
        k = self.covar_module.forward(self.train_x, self.train_x)
        n = k.shape[-1]
        return k + self.noise.to(k.device) * torch.eye(n, dtype=k.dtype, device=k.device)

    def _priors(self):
        total = 0.0
        mods = [self.covar_module] + ([self.covar_module.base_kernel] if hasattr(self.covar_module, "base_kernel") else [])
        for m in mods:
            for _, (prior, closure, _) in getattr(m, "_priors", {}).items():
                total = total + prior.log_prob(closure()).sum()
        if self.noise_prior is not None:
            total = total + self.noise_prior.log_prob(self.noise)
        return total

    def marginal_log_likelihood(self):
        """(log p(y | X) + log priors) / n, the quantity gpytorch's ExactMarginalLogLikelihood returns [3P]."""
        k = self._kxx()
        n = k.shape[-1]
        L = torch.linalg.cholesky(k)
        r = (self.train_y.to(k.device) - self.mean_constant.to(k.device)).unsqueeze(-1)
        alpha = torch.cholesky_solve(r, L)
        ll = -0.5 * (r * alpha).sum() - torch.log(torch.diagonal(L)).sum() - 0.5 * n * math.log(2 * math.pi)
        pri = self._priors()
        return (ll + (pri.to(ll.device) if torch.is_tensor(pri) else pri)) / n

    def invalidate(self):
        self._cache = None

    def _ensure_cache(self):
        """(L^-1, alpha, mean) of the fitted model; freezes the kernel hyper-parameters (prediction mode)."""
        if self._cache is None:
            with torch.no_grad():
                L = torch.linalg.cholesky(self._kxx())
                mu = self.mean_constant.detach().to(L.device)
                alpha = torch.cholesky_solve((self.train_y.to(L.device) - mu).unsqueeze(-1), L).squeeze(-1)
                Linv = torch.linalg.solve_triangular(L, torch.eye(L.shape[-1], dtype=L.dtype, device=L.device), upper=False)
            self._cache = (Linv, alpha, mu)
        for p in self.covar_module.parameters():
            p.requires_grad_(False)
        return self._cache

    def posterior(self, X):
        if X.dim() == 2:
            X = X.unsqueeze(-2)
        b = X.shape[0]
        Linv, alpha, mu = self._ensure_cache()
        xt = self.train_x.to(X.device).expand(b, *self.train_x.shape)
        ks = self.covar_module.forward(X, xt).squeeze(-2)
        kss = self.covar_module.forward(X, X).reshape(b)
        Li, ad = Linv.to(ks.device), alpha.to(ks.device)
        mean = mu.to(ks.device) + ks @ ad
        v = ks @ Li.transpose(-1, -2)
        return mean, kss - (v * v).sum(-1)


def fit_gpytorch_model(model, maxiter=200):
    """Maximise the marginal log likelihood (+ priors) over every trainable parameter with scipy L-BFGS-B, as
    botorch.fit_gpytorch_model does for the reference examples (examples/gabo_spd.py:194) [3P]."""
    import numpy as np
    from scipy.optimize import minimize
    params = [p for p in model.parameters()]
    for p in params:
        p.requires_grad_(True)
    shapes = [p.shape for p in params]
    sizes = [p.numel() for p in params]

    def set_params(v):
        off = 0
        with torch.no_grad():
            for p, sh, sz in zip(params, shapes, sizes):
                p.copy_(torch.as_tensor(v[off:off + sz], dtype=p.dtype).reshape(sh))
                off += sz

    def fun(v):
        set_params(v)
        for p in params:
            p.grad = None
        try:
            loss = -model.marginal_log_likelihood()
            loss.backward()
        except RuntimeError:          # a trial point outside the SPD cone of K + noise I
            return 1e10, np.zeros_like(v)
        g = np.concatenate([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().double().reshape(-1).numpy()
                            for p in params])
        return float(loss.item()), g

    x0 = np.concatenate([p.detach().cpu().double().reshape(-1).numpy() for p in params])
    res = minimize(fun, x0, jac=True, method="L-BFGS-B", options={"maxiter": maxiter})
    set_params(res.x)
    model.invalidate()
    return model
