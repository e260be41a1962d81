"""Fast path of the acquisition maximiser for the built-in surrogate: cost(x) = -acq(x) and its Euclidean gradient as a
fixed chain of HIP launches, without autograd:

    [matrix -> Mandel]  ->  kernel strip K(X*, X_train)  ->  gabo_gp_acquisition (posterior + EI/mean + d/dK*)
                        ->  kernel backward (d/dX*)      ->  [Mandel -> matrix]

It computes exactly what `acquisition_function(post_processing(x)[:, None])` and torch.autograd compute through
models.ExactGP / models.ExpectedImprovement (manifold_optimize.py:175-184 in the reference), in ~10 launches instead of
~150 tiny ones, which is what bounds the lock-step trust regions.  Anything it does not recognise (another surrogate,
another kernel, user pre/post-processing callables, second derivatives) stays on the generic autograd path.
"""
import math

import torch

from . import _lib, models, ops
from .kernel_utils import kernels_spd, kernels_sphere
from .Riemannian_utils import spd_utils_torch


def _surrogate_view(gp):
    """(base kernel, outputscale, constant mean, (L^-1, alpha), train_x) of a built-in surrogate in prediction mode, else None."""
    from ._compat import ScaleKernel
    if type(gp) is models.ExactGP:
        return gp.base_kernel, float(gp.outputscale), float(gp.mean), gp._train_cache(), gp.train_x
    if type(gp) is models.SingleTaskGP:
        cm = gp.covar_module
        base, outputscale = (cm.base_kernel, float(cm.outputscale.detach())) if type(cm) is ScaleKernel else (cm, 1.0)
        linv, alpha, mu = gp._ensure_cache()
        return base, outputscale, float(mu), (linv, alpha), gp.train_x
    return None


class FusedAcquisition:
    def __init__(self, acq, family, mode, beta, matrix_input, device, flavour="ai", view=None):
        base, outputscale, mean, (linv, alpha), train_x = view if view is not None else _surrogate_view(acq.model)
        self.family, self.mode, self.beta, self.matrix_input = family, mode, beta, matrix_input
        self.flavour = flavour          # SPD kernels: "ai" affine-invariant, "le" log-Euclidean, "frob" Frobenius
        self.kind = _lib.GABO_ACQ_EXPECTED_IMPROVEMENT if isinstance(acq, models.ExpectedImprovement) else _lib.GABO_ACQ_POSTERIOR_MEAN
        self.maximize = bool(acq.maximize)
        self.best_f = float(getattr(acq, "best_f", 0.0))
        self.mean, self.outputscale = mean, outputscale
        device = torch.device(device)
        on = lambda t_: t_ if (t_.device == device and t_.is_contiguous()) else t_.to(device).contiguous()    # noqa: E731
        self.linv = on(linv)
        # (gabo_gp_factor wrote L^-T next to L^-1: models.*._cache_linv_t, valid for exactly that cache object)
        held = getattr(acq.model, "_cache_linv_t", None)
        if held is not None and held[0] is getattr(acq.model, "_cache", None) and held[1].device == self.linv.device:
            self.linv_t = held[1]
        else:
            self.linv_t = self.linv.t().contiguous()
        # (... and the symmetric inverse A = L^-T L^-1, which the fused SPD kernels take in place of the two factors: models.*._cache_kinv)
        heldk = getattr(acq.model, "_cache_kinv", None)
        self.kinv = heldk[1] if (heldk is not None and heldk[0] is getattr(acq.model, "_cache", None) and heldk[1] is not None
                                 and heldk[1].device == self.linv.device) else None
        import os
        if os.environ.get("GABO_NO_KINV"):          # development A/B: the two triangular factors as in rounds 2-5
            self.kinv = None
        self.alpha = on(alpha)
        self.train = on(train_x)
        # d <= 12: value + gradient in ONE launch per evaluation (csrc/spd_acq.hip); the training side is factored once here
        self.single_launch = False
        if family == "spd" and flavour == "ai":
            d_spd = ops._mandel_dim(self.train.shape[-1])
            self.single_launch = (d_spd <= _lib.GABO_SPD_REG_MAX_DIM
                                  and self.train.shape[0] <= _lib.load().gabo_spd_acq_max_train(d_spd))
        if family == "sphere":
            # one launch per evaluation for the sphere kernels too (csrc/sphere_tr.hip); needs the training set in both layouts
            n_tr, dim_tr = self.train.shape
            self.single_launch = (7 * n_tr + 6 * dim_tr) * 8 <= 150 * 1024 and n_tr <= 4096 and 2 <= dim_tr <= 512
            self.train_t = self.train.t().contiguous() if self.single_launch else None
        self.metric = {"ai": _lib.GABO_METRIC_AFFINE_INVARIANT, "le": _lib.GABO_METRIC_LOG_EUCLIDEAN, "frob": _lib.GABO_METRIC_FROBENIUS}[flavour]
        self.train_factors = None
        if self.single_launch and family == "spd":
            # (gabo_spd_gp_prepare wrote them next to the factor: models.ExactGP._cache_factors, valid for exactly that cache object)
            heldf = getattr(acq.model, "_cache_factors", None)
            if heldf is not None and heldf[0] is getattr(acq.model, "_cache", None) and heldf[1] is not None and heldf[1].device == self.linv.device:
                self.train_factors = heldf[1]
            else:
                self.train_factors = ops.spd_acq_prepare_train(self.train)
        if family == "spd" and flavour != "ai":
            # ||0 + 1e-15||_F^2 = d^2 1e-30 (spd_utils_torch.py:156): k(x, x) = 1 to the last bit; logm of the training set once
            self.kxx = 1.0
            self.train_feat = ops.spd_logm_mandel(self.train) if flavour == "le" else self.train
            d_spd = ops._mandel_dim(self.train.shape[-1])
            if (d_spd <= 8 and self.train.shape[0] <= _lib.load().gabo_spd_acq_max_train(d_spd) and mode == _lib.GABO_OUT_GAUSSIAN):
                # one launch per evaluation for these too: the "factors" are the training features, entry-major
                self.single_launch = True
                self.train_factors = self.train_feat.t().contiguous()
        elif family == "spd":
            # d(X, X)^2 = 1e-15 exactly (the eigenvalues of L^-1 X L^-T are 1 to rounding): spd_utils_torch.py:120
            self.kxx = math.exp(-beta * (1e-15 if mode == _lib.GABO_OUT_GAUSSIAN else math.sqrt(1e-15)))
        else:
            # <x, x> is clamped to 1 - 1e-15 (sphere_utils_torch.py:53): d(x, x) = acos(1 - 1e-15)
            t0 = math.acos(1.0 - 1e-15)
            self.kxx = math.exp(-beta * (t0 * t0 if mode == _lib.GABO_OUT_GAUSSIAN else t0))

    @staticmethod
    def build(acq, post_processing, device):
        """-> FusedAcquisition, or None when the acquisition / surrogate / kernel / post-processing is not a built-in.
        Cached on the acquisition object (one sweep asks twice: raw-sample scoring and the solve); the key includes the
        surrogate's prediction cache, so refitting or changing the data rebuilds it."""
        gp = getattr(acq, "model", None)
        cached = getattr(acq, "_gabo_fused", None)
        if cached is not None:
            post_c, dev_c, cache_c, best_c, max_c, fused_c = cached      # (the objects themselves are held: no id() reuse)
            if (post_c is post_processing and dev_c == str(device) and cache_c is not None and cache_c is getattr(gp, "_cache", None)
                    and best_c == getattr(acq, "best_f", None) and max_c == getattr(acq, "maximize", None)):
                return fused_c
        fused = FusedAcquisition._build(acq, post_processing, device)
        if fused is not None and getattr(gp, "_cache", None) is not None:
            try:
                object.__setattr__(acq, "_gabo_fused", (post_processing, str(device), gp._cache, getattr(acq, "best_f", None),
                                                        getattr(acq, "maximize", None), fused))
            except Exception:       # noqa: BLE001
                pass
        return fused

    @staticmethod
    def _build(acq, post_processing, device):
        if not isinstance(acq, (models.ExpectedImprovement, models.PosteriorMean)):
            return None
        view = _surrogate_view(getattr(acq, "model", None))
        if view is None:
            return None
        k = view[0]
        if type(k) in (kernels_spd.SpdAffineInvariantGaussianKernel, kernels_spd.SpdAffineInvariantLaplaceKernel):
            if post_processing is not spd_utils_torch.symmetric_matrix_to_vector_mandel_torch:
                return None
            mode = _lib.GABO_OUT_GAUSSIAN if type(k) is kernels_spd.SpdAffineInvariantGaussianKernel else _lib.GABO_OUT_LAPLACE
            return FusedAcquisition(acq, "spd", mode, k.beta_float(), True, device, view=view)
        if type(k) in (kernels_spd.SpdLogEuclideanGaussianKernel, kernels_spd.SpdFrobeniusGaussianKernel):
            if post_processing is not spd_utils_torch.symmetric_matrix_to_vector_mandel_torch:
                return None
            ls = float(k.lengthscale.detach().double())
            flavour = "le" if type(k) is kernels_spd.SpdLogEuclideanGaussianKernel else "frob"
            return FusedAcquisition(acq, "spd", _lib.GABO_OUT_GAUSSIAN, 1.0 / (ls * ls), True, device, flavour=flavour)
        if type(k) in (kernels_sphere.SphereGaussianKernel, kernels_sphere.SphereLaplaceKernel):
            if post_processing is not None:
                return None
            if type(k) is kernels_sphere.SphereGaussianKernel:
                mode, beta = _lib.GABO_OUT_GAUSSIAN, float(k.beta.double())
            else:
                ls = float(k.lengthscale.double())
                mode, beta = _lib.GABO_OUT_LAPLACE, 1.0 / (ls * ls)
            return FusedAcquisition(acq, "sphere", mode, beta, False, device)
        return None

    def _strip(self, x):
        pts = ops.matrix_to_mandel(x) if self.matrix_input else x.contiguous()
        if self.family == "spd" and self.flavour == "ai":
            return pts, ops.spd_ai_pairwise(pts, self.train, self.beta, self.mode), None
        if self.family == "spd":
            feat = ops.spd_logm_mandel(pts) if self.flavour == "le" else pts
            return pts, ops.frobenius_pairwise(feat, self.train_feat, self.beta, self.mode), feat
        c = pts @ self.train.t()
        return pts, ops.sphere_from_inner(c, self.beta, self.mode, 0), c

    def egrad_mandel(self, pts, active_ptr=None, out=None):
        """Euclidean gradient of the cost at SPD points given (and returned) as Mandel vectors: the chain without the two
        Mandel maps, used by the fused trust-region inner loop."""
        if self.single_launch:
            return self._single(pts, True, active_ptr, out)[1]
        _, ks, feat = self._strip_mandel(pts)
        _, gk = ops.gp_acquisition(ks, self.alpha, self.linv, self.linv_t, self.mean, self.outputscale, self.kxx, self.best_f,
                                   self.kind, self.maximize, out_sign=-1.0, need_grad=True)
        return self._strip_backward(pts, feat, gk)

    def _strip_mandel(self, pts):
        if self.flavour == "ai":
            return pts, ops.spd_ai_pairwise(pts, self.train, self.beta, self.mode), None
        feat = ops.spd_logm_mandel(pts) if self.flavour == "le" else pts
        return pts, ops.frobenius_pairwise(feat, self.train_feat, self.beta, self.mode), feat

    def _strip_backward(self, pts, feat, gk):
        """d/d pts (Mandel) of sum(gk * strip)"""
        if self.flavour == "ai":
            return ops.spd_ai_backward(pts, self.train, gk, self.beta, self.mode, wrt=1)
        g = ops.frobenius_backward(feat, self.train_feat, gk, self.beta, self.mode, wrt=1)
        return ops.spd_logm_mandel_backward(pts, g) if self.flavour == "le" else g

    def sphere_acq_params(self):
        """The surrogate as the gabo_sphere_acq_params struct of the C ABI."""
        # (A = L^-T L^-1 for both factor slots when the model's cache holds it: half the matrix-vector products of an evaluation, csrc/sphere_tr.hip)
        la, lb = (self.kinv.data_ptr(), self.kinv.data_ptr()) if self.kinv is not None else (self.linv.data_ptr(), self.linv_t.data_ptr())
        return _lib.SphereAcqParams(self.train.data_ptr(), self.train_t.data_ptr(), self.alpha.data_ptr(), la,
                                    lb, self.train.shape[0], self.train.shape[1], self.beta, int(self.mode), self.mean,
                                    self.outputscale, self.kxx, self.best_f, int(self.kind), 1 if self.maximize else 0, -1.0)

    def acq_params(self):
        """The surrogate as the gabo_spd_acq_params struct of the C ABI (single-launch path only)."""
        # (A for both factor slots when the model's cache holds it: one matrix-vector product per evaluation instead of two, csrc/spd_acq_body.hpp)
        la, lb = (self.kinv.data_ptr(), self.kinv.data_ptr()) if self.kinv is not None else (self.linv.data_ptr(), self.linv_t.data_ptr())
        return _lib.AcqParams(self.train_factors.data_ptr(), self.alpha.data_ptr(), la, lb,
                              self.train.shape[0], self.beta, int(self.mode) | self.metric, self.mean, self.outputscale, self.kxx, self.best_f,
                              int(self.kind), 1 if self.maximize else 0, -1.0)

    def _single(self, pts, need_grad, active_ptr=None, out=None):
        la, lb = (self.kinv, self.kinv) if self.kinv is not None else (self.linv, self.linv_t)
        return ops.spd_acq_eval(pts, self.train_factors, self.alpha, la, lb, self.beta, int(self.mode) | self.metric, self.mean,
                                self.outputscale, self.kxx, self.best_f, self.kind, self.maximize, out_sign=-1.0, need_grad=need_grad,
                                active_ptr=active_ptr, out=out)

    def cost(self, x):
        if self.single_launch and self.family == "sphere":
            return ops.sphere_acq_eval(x.detach(), self.sphere_acq_params(), need_grad=False)[0]
        if self.single_launch:
            return self._single(ops.matrix_to_mandel(x.detach()) if self.matrix_input else x.detach(), False)[0]
        _, ks, _ = self._strip(x.detach())
        val, _ = ops.gp_acquisition(ks, self.alpha, self.linv, self.linv_t, self.mean, self.outputscale, self.kxx, self.best_f,
                                    self.kind, self.maximize, out_sign=-1.0, need_grad=False)
        return val

    def cost_egrad(self, x):
        if self.single_launch and self.family == "sphere":
            return ops.sphere_acq_eval(x.detach(), self.sphere_acq_params(), need_grad=True)
        if self.single_launch:
            val, g = self._single(ops.matrix_to_mandel(x.detach()) if self.matrix_input else x.detach(), True)
            return val, (ops.mandel_to_matrix(g) if self.matrix_input else g)
        pts, ks, c = self._strip(x.detach())
        val, gk = ops.gp_acquisition(ks, self.alpha, self.linv, self.linv_t, self.mean, self.outputscale, self.kxx, self.best_f,
                                     self.kind, self.maximize, out_sign=-1.0, need_grad=True)
        if self.family == "spd":
            g = self._strip_backward(pts, c, gk)         # (c = the per-point features of the log-Euclidean / Frobenius flavours)
        else:
            g = (gk * ops.sphere_from_inner(c, self.beta, self.mode, 1)) @ self.train
        return val, (ops.mandel_to_matrix(g) if self.matrix_input else g)
