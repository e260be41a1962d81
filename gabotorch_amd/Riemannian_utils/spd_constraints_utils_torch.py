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


def builtin_constraint(con):
    """(kind, bound) when `con` is one of the two constraints above with its bound bound by functools.partial - the way the
    reference examples build them (examples/gabo_spd.py:136-138) - so that the library can evaluate it on the device; else None."""
    import functools
    if not isinstance(con, functools.partial) or con.func not in (max_eigenvalue_constraint_torch, min_eigenvalue_constraint_torch):
        return None
    name = "maximum_eigenvalue" if con.func is max_eigenvalue_constraint_torch else "minimum_eigenvalue"
    if len(con.args) == 1 and not con.keywords:
        bound = con.args[0]
    elif not con.args and set(con.keywords) == {name}:
        bound = con.keywords[name]
    else:
        return None
    if torch.is_tensor(bound):
        if bound.numel() != 1:
            return None
        bound = bound.item()
    return (_lib.GABO_CONSTRAINT_MAX_EIGENVALUE if con.func is max_eigenvalue_constraint_torch else _lib.GABO_CONSTRAINT_MIN_EIGENVALUE,
            float(bound))
