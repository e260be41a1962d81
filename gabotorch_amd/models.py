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

from . import _lib as _l, ops as _ops


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
            n = self.train_x.shape[-2]
            if self.train_x.is_cuda and 0 < n <= _l.GABO_GP_FACTOR_MAX_N:
                # one launch instead of Cholesky (+ its info read-back), cholesky_solve and a triangular solve (csrc/gp_factor.hip)
                def drop(model=self):
                    # (the factor this cache was to hold does not exist: ops.check_deferred calls this before it raises)
                    model._cache = None
                    model._cache_linv_t = None
                    model._cache_factors = None
                    model._cache_kinv = None
                with torch.no_grad():
                    from .kernel_utils import kernels_spd as _ks
                    bk = self.base_kernel
                    if type(bk) in (_ks.SpdAffineInvariantGaussianKernel, _ks.SpdAffineInvariantLaplaceKernel) and self.train_x.dim() == 2 \
                            and self.train_x.shape[-1] <= _l.GABO_SPD_REG_MAX_DIM * (_l.GABO_SPD_REG_MAX_DIM + 1) // 2:
                        # (what bk.forward launches - the x1-is-x2 build - then the factor and the fused evaluators' training factors: one host call)
                        mode = _l.GABO_OUT_GAUSSIAN if type(bk) is _ks.SpdAffineInvariantGaussianKernel else _l.GABO_OUT_LAPLACE
                        Linv, Linv_t, alpha, factors, kinv = _ops.spd_gp_prepare(self.train_x, self.train_y, bk.beta_float(), mode, float(self.outputscale),
                                                                                 float(self.noise), float(self.mean), on_fail=drop)
                        self._cache = (Linv, alpha)
                        self._cache_linv_t = (self._cache, Linv_t)
                        self._cache_factors = (self._cache, factors)
                        self._cache_kinv = (self._cache, kinv)
                        return self._cache
                    kb = bk.forward(self.train_x, self.train_x).double()
                    Linv, Linv_t, alpha, kinv = _ops.gp_factor(kb, self.train_y, float(self.outputscale), float(self.noise), float(self.mean), defer_check=True,
                                                               on_fail=drop, want_kinv=True)
                self._cache = (Linv, alpha)
                self._cache_linv_t = (self._cache, Linv_t)
                self._cache_kinv = (self._cache, kinv)
                return self._cache
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
        from . import ops as _ops
        _ops.check_deferred()
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


def initialize_q_batch(X, Y, n, eta=1.0, generator=None):
    """botorch.optim.initializers.initialize_q_batch [3P]: Boltzmann sampling on standardised values, argmax forced in.
    `generator` (not in botorch): the random stream of the selection, so that every rank of a sharded sweep selects the same rows."""
    n_samples = X.shape[0]
    if n > n_samples:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({n_samples})")
    if n == n_samples:
        return X
    Ystd = Y.std()
    if Ystd == 0:
        warnings.warn("All acquisition values for raw samples points are the same.", BadInitialCandidatesWarning)
        return X[torch.randperm(n=n_samples, device=X.device, generator=generator)][:n]
    max_val, max_idx = torch.max(Y, dim=0)
    Z = (Y - Y.mean()) / Ystd
    etaZ = eta * Z
    weights = torch.exp(etaZ)
    while torch.isinf(weights).any():
        etaZ *= 0.5
        weights = torch.exp(etaZ)
    idcs = torch.multinomial(weights, n, generator=generator)
    if max_idx not in idcs:
        idcs[-1] = max_idx
    return X[idcs]


def initialize_q_batch_nonneg(X, Y, n, eta=1.0, alpha=1e-4, generator=None):
    """botorch.optim.initializers.initialize_q_batch_nonneg [3P]; `generator` as in initialize_q_batch."""
    n_samples = X.shape[0]
    if n > n_samples:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({n_samples})")
    if n == n_samples:
        return X
    max_val, max_idx = torch.max(Y, dim=0)
    if torch.any(max_val <= 0):
        warnings.warn("All acquisition values for raw sampled points are nonpositive, so initial conditions are being "
                      "selected randomly.", BadInitialCandidatesWarning)
        return X[torch.randperm(n=n_samples, device=X.device, generator=generator)][:n]
    pos = Y > 0
    num_pos = int(pos.sum().item())
    if num_pos < n:
        remaining = n - num_pos
        rand_idx = torch.randperm(n_samples - num_pos, device=Y.device, generator=generator)[:remaining]
        xpos, xneg = X[pos], X[~pos][rand_idx]
        return torch.cat([xpos, xneg], dim=0)[torch.randperm(n, device=X.device, generator=generator)]
    alpha_pos = Y >= alpha * max_val
    while alpha_pos.sum() < n:
        alpha = 0.1 * alpha
        alpha_pos = Y >= alpha * max_val
    alpha_pos_idcs = torch.arange(len(Y), device=Y.device)[alpha_pos]
    weights = torch.exp(eta * (Y[alpha_pos] / max_val - 1))
    idcs = alpha_pos_idcs[torch.multinomial(weights, n, generator=generator)]
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


class _ExactMll(torch.autograd.Function):
    """log N(y | mean, outputscale * K + noise * I) for a base Gram matrix K on the device: forward and the whole gradient in one
    gabo_gp_mll_gram launch (d ll / d K = outputscale * W / 2 with W = alpha alpha^T - Ky^-1).  A Gram matrix that is not positive
    definite gives -inf (and zero gradients) instead of an exception: the fit's line searches then reject the point."""

    @staticmethod
    def forward(ctx, kbase, y, outputscale, noise, mean):
        from . import ops
        out, w = ops.gp_mll_gram(kbase.detach(), y.detach(), outputscale.item(), noise.item(), mean.item())
        ctx.save_for_backward(out, w)
        ctx.os = outputscale.item()
        ctx.host = [(t.device, t.dtype) for t in (outputscale, noise, mean)]
        return torch.where(out[5] > 0, torch.full_like(out[0], -math.inf), out[0])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        out, w = ctx.saved_tensors
        ok = (out[5] == 0).to(out.dtype)
        scalars = (g * ok * out[2:5])
        gk = (g * ok * 0.5 * ctx.os) * w
        if any(dev_.type == "cpu" for dev_, _ in ctx.host):
            scalars = scalars.cpu()                       # one read-back for the three host-resident hyper-parameters
        return (gk, None) + tuple(scalars[i].to(dev_, dt_) for i, (dev_, dt_) in enumerate(ctx.host))


class SingleTaskGP(torch.nn.Module):
    """Constant-mean exact GP with a Gaussian likelihood and trainable hyper-parameters, laid out like the models of the
    reference examples (examples/gabo_spd.py:165-176: ScaleKernel(base, outputscale_prior=Gamma(2, .15)), noise prior
    Gamma(1.1, .05), noise constraint GreaterThan(1e-8), initial noise = prior mode).  Priors registered on the kernels through
    the gpytorch-compatible `register_prior` are picked up when the stand-in Kernel base class is in use."""

    def __init__(self, train_x, train_y, covar_module, noise_prior=None, noise_lower_bound=1e-8, initial_noise=None):
        super().__init__()
        self.train_x = train_x.double()
        self.train_y = train_y.double().reshape(-1)
        self.covar_module = covar_module
        self.noise_prior = noise_prior
        self.noise_lower_bound = float(noise_lower_bound)
        if initial_noise is not None and initial_noise > noise_lower_bound:
            init_noise = float(initial_noise)
        else:
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
        """(log p(y | X) + log priors) / n, the quantity gpytorch's ExactMarginalLogLikelihood returns [3P].
        Up to GABO_GP_MLL_LARGE_MAX_N training points the likelihood and its gradient with respect to the Gram matrix, outputscale, noise
        and mean are ONE launch (gabo_gp_mll_gram) behind a custom autograd node, so autograd only has to differentiate the kernel
        itself (its own HIP backward); above that, torch's Cholesky / solve."""
        from . import _lib
        cm = self.covar_module
        scaled = hasattr(cm, "base_kernel") and type(cm).__name__ == "ScaleKernel" and cm.raw_outputscale.numel() == 1
        kb = (cm.base_kernel if scaled else cm).forward(self.train_x, self.train_x)
        if kb.is_cuda and kb.dim() == 2 and kb.shape[-1] <= _lib.GABO_GP_MLL_LARGE_MAX_N:
            os_ = cm.outputscale.double().reshape(()) if scaled else torch.ones((), dtype=torch.float64)
            ll = _ExactMll.apply(kb.double(), self.train_y.to(kb.device), os_, self.noise.reshape(()), self.mean_constant.reshape(()))
            pri = self._priors()
            return (ll + (pri.to(ll.device) if torch.is_tensor(pri) else pri)) / kb.shape[-1]
        return self._marginal_log_likelihood_torch()

    def _marginal_log_likelihood_torch(self):
        """The same through torch.linalg (any size; the path the launch above is tested against)."""
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

    def set_train_data(self, inputs=None, targets=None, strict=True):
        """gpytorch ExactGP.set_train_data [3P]: replace the training set between BO iterations (examples/gabo_spd.py:216; the
        hyper-parameters keep their current values and stay trainable for the next fit)"""
        if inputs is not None:
            if strict and tuple(inputs.shape) != tuple(self.train_x.shape):
                raise RuntimeError("Cannot modify shape of inputs (expected strict=False)")
            self.train_x = inputs.double()
        if targets is not None:
            if strict and targets.numel() != self.train_y.numel():
                raise RuntimeError("Cannot modify shape of targets (expected strict=False)")
            self.train_y = targets.double().reshape(-1)
        self.invalidate()

    # ---- fast surrogate fit: the distances are evaluated once, each evaluation is one gabo_gp_mll launch ------------------
    def _stationary_form(self):
        """(E, theta_fn, outputscale_fn) when the covariance is [ScaleKernel of] one of the path's plain kernels, all of which
        are exp(-theta * E) with E = d^2 or d fixed during a fit; None for anything else (nested kernels carry extra
        parameters inside the distance, batched hyper-parameters, more than GABO_GP_MLL_LARGE_MAX_N points)."""
        from . import _lib, ops
        from .kernel_utils import kernels_sphere as ksph
        from .kernel_utils import kernels_spd as kspd
        cm = self.covar_module
        base = getattr(cm, "base_kernel", None)
        if base is None:
            base, os_fn = cm, (lambda: torch.ones((), dtype=torch.float64))
        elif type(cm).__name__ == "ScaleKernel" and cm.raw_outputscale.numel() == 1:
            os_fn = lambda: cm.outputscale.double().reshape(())                                # noqa: E731
        else:
            return None
        x = self.train_x
        if x.dim() != 2 or not 1 <= x.shape[0] <= _lib.GABO_GP_MLL_LARGE_MAX_N:
            return None
        kind = type(base)
        dist_mode = _lib.GABO_OUT_DISTANCE
        beta_fn = lambda: base.beta.double().reshape(())                                       # noqa: E731
        ls_fn = lambda: 1.0 / base.lengthscale.double().reshape(()) ** 2                       # noqa: E731
        if kind in (kspd.SpdAffineInvariantGaussianKernel, kspd.SpdAffineInvariantLaplaceKernel):
            dist, theta_fn = (lambda v: ops.spd_ai_pairwise(v, v, 1.0, dist_mode)), beta_fn
            power = 2 if kind is kspd.SpdAffineInvariantGaussianKernel else 1
        elif kind is kspd.SpdFrobeniusGaussianKernel:
            dist, theta_fn, power = (lambda v: ops.frobenius_pairwise(v, v, 1.0, dist_mode)), ls_fn, 2
        elif kind is kspd.SpdLogEuclideanGaussianKernel:
            dist, theta_fn, power = (lambda v: ops.frobenius_pairwise(*(2 * (ops.spd_logm_mandel(v),)), 1.0, dist_mode)), ls_fn, 2
        elif kind is ksph.SphereGaussianKernel:
            dist, theta_fn, power = (lambda v: ops.sphere_pairwise(v, v, 1.0, dist_mode)), beta_fn, 2
        elif kind is ksph.SphereLaplaceKernel:
            dist, theta_fn, power = (lambda v: ops.sphere_pairwise(v, v, 1.0, dist_mode)), ls_fn, 1
        else:
            return None
        if theta_fn().numel() != 1:
            return None
        with torch.no_grad():
            d = dist(x.to(ops._device_for(x)))
            e = (d * d if power == 2 else d).contiguous()
        return e, theta_fn, os_fn

    def _fast_mll_closure(self):
        """() -> (value, backward_fn) evaluating `marginal_log_likelihood` and the gradients of all parameters with one
        gabo_gp_mll launch: the device returns d ll / d (theta, outputscale, noise, mean); the chain rule through the parameter
        transforms and the priors is a handful of scalar torch operations on the parameters' own device."""
        from . import ops
        form = self._stationary_form()
        if form is None:
            return None
        e, theta_fn, os_fn = form
        y = self.train_y.to(e.device).contiguous()
        n = y.numel()

        def evaluate():
            theta, os_, noise, mean = theta_fn(), os_fn(), self.noise, self.mean_constant
            ll, g_theta, g_os, g_noise, g_mean, bad = ops.gp_mll(e, y, theta.item(), os_.item(), noise.item(), mean.item())
            if bad:
                raise torch.linalg.LinAlgError("K + noise I is not positive definite")
            pri = self._priors()
            value = (ll + (pri.item() if torch.is_tensor(pri) else pri)) / n
            # a linear stand-in with the same first derivatives as ll at this point, so that autograd does the chain rule
            proxy = (g_theta * theta + g_os * os_.to(theta.device) + g_noise * noise.to(theta.device)
                     + g_mean * mean.to(theta.device) + (pri.to(theta.device) if torch.is_tensor(pri) else 0.0)) / n
            return value, proxy

        return evaluate

    def _fast_scalar_objective(self, params, evaluator=None):
        """v -> (loss, gradient) in plain Python floats for the layout every reference example uses (stand-in ScaleKernel / kernel
        classes with softplus constraints, Gamma priors): the chain rule through softplus and the priors costs microseconds, so an
        L-BFGS evaluation is the gabo_gp_mll launch and its 48-byte read-back.  None when the model is laid out differently (the
        caller then lets autograd do the chain rule).
        evaluator(theta, outputscale, noise, mean) -> (ll, d theta, d outputscale, d noise, d mean, not_pd): replaces the launch on the
        fixed distance matrix for kernels with further parameters inside the distance (the nested kernels: manifold_gp_fit.py); `params`
        then lists the SCALAR hyper-parameters only."""
        import numpy as np

        from . import _compat, ops
        if _compat.HAVE_GPYTORCH:
            return None
        if evaluator is None:
            form = self._stationary_form()
            if form is None:
                return None
            e = form[0]
        cm = self.covar_module
        base = getattr(cm, "base_kernel", cm)
        index = {id(p): i for i, p in enumerate(params)}
        plan = {}                      # hyper-parameter -> (position in v, fp32?, lower bound)

        def softplus_param(name, module, raw_name):
            raw = getattr(module, raw_name)
            con = getattr(module, raw_name + "_constraint", None)
            if type(con) not in (_compat.GreaterThan, _compat.Positive) or raw.numel() != 1 or id(raw) not in index:
                return False
            plan[name] = (index[id(raw)], raw.dtype == torch.float32, float(con.lower_bound))
            return True

        uses_beta = hasattr(base, "raw_beta")
        if not softplus_param("theta", base, "raw_beta" if uses_beta else "raw_lengthscale"):
            return None
        if base is not cm and not softplus_param("os", cm, "raw_outputscale"):
            return None
        if id(self.raw_noise) not in index or id(self.mean_constant) not in index:
            return None
        plan["noise"] = (index[id(self.raw_noise)], False, self.noise_lower_bound)
        i_mean = index[id(self.mean_constant)]
        if len(plan) + 1 != len(params):
            return None
        priors = []                    # (hyper-parameter the prior is on, concentration, rate)
        on_base = {"beta_prior": "theta", "lengthscale_prior": "theta"}      # (a lengthscale prior is on l, the constrained value)
        for module, allowed in ([(cm, on_base)] if base is cm else [(cm, {"outputscale_prior": "os"}), (base, on_base)]):
            for name, (prior, _, _) in getattr(module, "_priors", {}).items():
                if name not in allowed or type(prior) is not GammaPrior:
                    return None
                priors.append((allowed[name], prior.concentration, prior.rate))
        if self.noise_prior is not None:
            if type(self.noise_prior) is not GammaPrior:
                return None
            priors.append(("noise", self.noise_prior.concentration, self.noise_prior.rate))
        n = self.train_y.numel()
        if evaluator is None:
            y = self.train_y.to(e.device).contiguous()
            evaluator = lambda theta, os_, noise, mean: ops.gp_mll(e, y, theta, os_, noise, mean)      # noqa: E731

        def objective(v):
            val, dval = {}, {}
            for name, (i, is32, lb) in plan.items():
                raw = float(np.float32(v[i])) if is32 else float(v[i])
                val[name] = lb + max(raw, 0.0) + math.log1p(math.exp(-abs(raw)))           # softplus
                dval[name] = 1.0 / (1.0 + math.exp(-raw)) if raw >= 0 else math.exp(raw) / (1.0 + math.exp(raw))
            positive = val["theta"]                       # beta, or the lengthscale l with theta = l^-2
            theta, dtheta = (positive, 1.0) if uses_beta else (positive ** -2, -2.0 * positive ** -3)
            ll, g_theta, g_os, g_noise, g_mean, bad = evaluator(theta, val.get("os", 1.0), val["noise"], float(v[i_mean]))
            grad = np.zeros_like(v)
            if bad:
                return 1e10, grad
            d_positive = {"theta": g_theta * dtheta, "os": g_os, "noise": g_noise}       # d total / d (constrained value)
            total = ll
            for name, a, b in priors:
                x = val[name]
                total += a * math.log(b) + (a - 1.0) * math.log(x) - b * x - math.lgamma(a)
                d_positive[name] += (a - 1.0) / x - b
            for name, (i, _, _) in plan.items():
                grad[i] = -d_positive[name] * dval[name] / n
            grad[i_mean] = -g_mean / n
            return -total / n, grad

        return objective

    def _ensure_cache(self):
        """(L^-1, alpha, mean) of the fitted model; freezes the kernel hyper-parameters (prediction mode)."""
        if self._cache is None:
            self._factor_in_one_launch()
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

    def _factor_in_one_launch(self):
        """gabo_gp_factor for the plain ScaleKernel(base) / base covariance modules on a HIP device (csrc/gp_factor.hip); leaves the cache
        empty otherwise (the torch route below then fills it)."""
        from . import _compat, _lib as _l, ops as _ops
        cm = self.covar_module
        n = self.train_x.shape[-2]
        if not (self.train_x.is_cuda and 0 < n <= _l.GABO_GP_FACTOR_MAX_N):
            return
        if type(cm) is _compat.ScaleKernel and not _compat.HAVE_GPYTORCH:
            base, outputscale = cm.base_kernel, float(cm.outputscale.detach())
        else:
            return
        def drop(model=self):
            model._cache = None
            model._cache_linv_t = None
            model._cache_kinv = None
            model._cache_factors = None
        from .kernel_utils import kernels_spd as _ks
        if type(base) in (_ks.SpdAffineInvariantGaussianKernel, _ks.SpdAffineInvariantLaplaceKernel) and self.train_x.dim() == 2 \
                and self.train_x.shape[-1] <= _l.GABO_SPD_REG_MAX_DIM * (_l.GABO_SPD_REG_MAX_DIM + 1) // 2:
            # (as models.ExactGP: Gram of the training set, factor, symmetric inverse and the fused evaluators' training factors from ONE host call)
            with torch.no_grad():
                mu = self.mean_constant.detach().to(self.train_x.device)
                mode = _l.GABO_OUT_GAUSSIAN if type(base) is _ks.SpdAffineInvariantGaussianKernel else _l.GABO_OUT_LAPLACE
                Linv, Linv_t, alpha, factors, kinv = _ops.spd_gp_prepare(self.train_x, self.train_y, base.beta_float(), mode, outputscale,
                                                                         float(self.noise.detach()), float(mu), on_fail=drop)
            self._cache = (Linv, alpha, mu)
            self._cache_linv_t = (self._cache, Linv_t)
            self._cache_kinv = (self._cache, kinv)
            self._cache_factors = (self._cache, factors)
            return
        with torch.no_grad():
            kb = base.forward(self.train_x, self.train_x)
            if kb.dim() != 2:
                return
            mu = self.mean_constant.detach().to(kb.device)
            Linv, Linv_t, alpha, kinv = _ops.gp_factor(kb.double(), self.train_y, outputscale, float(self.noise.detach()), float(mu), defer_check=True,
                                                       on_fail=drop, want_kinv=True)
        self._cache = (Linv, alpha, mu)
        self._cache_linv_t = (self._cache, Linv_t)
        self._cache_kinv = (self._cache, kinv)

    def posterior(self, X):
        if X.dim() == 2:
            X = X.unsqueeze(-2)
        b = X.shape[0]
        Linv, alpha, mu = self._ensure_cache()
        from . import ops as _ops
        _ops.check_deferred()
        xt = self.train_x.to(X.device).expand(b, *self.train_x.shape)
        ks = self.covar_module.forward(X, xt).squeeze(-2)
        kss = self.covar_module.forward(X, X).reshape(b)
        Li, ad = Linv.to(ks.device), alpha.to(ks.device)
        mean = mu.to(ks.device) + ks @ ad
        v = ks @ Li.transpose(-1, -2)
        return mean, kss - (v * v).sum(-1)


def fit_gpytorch_model(model, maxiter=200, fast=True):
    """Maximise the marginal log likelihood (+ priors) over every trainable parameter with scipy L-BFGS-B, as
    botorch.fit_gpytorch_model does for the reference examples (examples/gabo_spd.py:194) [3P].
    fast=True: for the plain kernels of the path the pairwise distances are evaluated once and every L-BFGS evaluation is a single
    gabo_gp_mll launch (value + analytic gradient) instead of a kernel launch, a Cholesky and their autograd; fast=False (and any
    other model) differentiates `marginal_log_likelihood` with autograd."""
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

    scalar = model._fast_scalar_objective(params) if fast and hasattr(model, "_fast_scalar_objective") else None
    fast = model._fast_mll_closure() if fast and scalar is None and hasattr(model, "_fast_mll_closure") else None

    def fun(v):
        if scalar is not None:
            return scalar(v)
        set_params(v)
        for p in params:
            p.grad = None
        try:
            if fast is not None:
                value, proxy = fast()
                (-proxy).backward()
                loss = torch.tensor(-value)
            else:
                loss = -model.marginal_log_likelihood()
                loss.backward()
        except torch.linalg.LinAlgError:      # a trial point outside the SPD cone of K + noise I (torch.linalg.cholesky); any other
            return 1e10, np.zeros_like(v)      # error - a failed launch, a non-SPD kernel INPUT - is a real failure and propagates
        g = np.concatenate([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().double().reshape(-1).numpy()
                            for p in params])
        value = float(loss.item())
        if not math.isfinite(value):  # (the one-launch likelihood reports a matrix that is not positive definite as -inf)
            return 1e10, np.zeros_like(v)
        return value, g

    x0 = np.concatenate([p.detach().cpu().double().reshape(-1).numpy() for p in params])
    res = minimize(fun, x0, jac=True, method="L-BFGS-B", options={"maxiter": maxiter})
    set_params(res.x)
    model.invalidate()
    return model
