"""Eigenvalue inequality constraints with the reference's names (BoManifolds/Riemannian_utils/spd_constraints_utils_torch.py:17-50).
Value and gradient come from one batched HIP eigen-decomposition (gradient of an extreme eigenvalue = v v^T)."""
import torch

from .. import _lib, ops


class _ExtremeEig(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        if not x.requires_grad:            # value only (e.g. the strict solver's feasibility test): skip the v v^T output
            return ops.spd_manifold_op(op, x).to(x.dtype)
        lam, vv = ops.spd_manifold_op(op, x, want_grad=True)
        ctx.save_for_backward(vv)
        return lam.to(x.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (vv,) = ctx.saved_tensors
        return g[..., None, None] * vv.to(g.dtype), None


def max_eigenvalue_constraint_torch(x, maximum_eigenvalue):
    """maximum_eigenvalue - lambda_max(x)  (>= 0 when satisfied); x (..., d, d)."""
    return maximum_eigenvalue - _ExtremeEig.apply(x, _lib.GABO_SPD_EIGMAX)


def min_eigenvalue_constraint_torch(x, minimum_eigenvalue):
    """lambda_min(x) - minimum_eigenvalue."""
    return _ExtremeEig.apply(x, _lib.GABO_SPD_EIGMIN) - minimum_eigenvalue
