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
    """(kind, bound[, lift]) when `con` is a constraint the library can evaluate inside its trust-region kernel, else None:
    one of the two constraints above with its bound bound by functools.partial - the way the reference examples build them
    (examples/gabo_spd.py:136-138) - or max/min_eigenvalue_nested_spd_constraint with the bound and the mapping (W, V, C, K) bound by
    keyword (examples/hd_bo_spd/benchmark_examples/hd_gabo_spd.py:244-257); lift = the mapping tensors, for the caller to compare and
    hand to ops.nested_spd_lift_prepare."""
    import functools

    from ..nested_mappings import nested_spd_constraints_utils as nested
    if not isinstance(con, functools.partial):
        return None
    if con.func in (nested.max_eigenvalue_nested_spd_constraint, nested.min_eigenvalue_nested_spd_constraint):
        is_max = con.func is nested.max_eigenvalue_nested_spd_constraint
        name = "maximum_eigenvalue" if is_max else "minimum_eigenvalue"
        mapping = ("projection_matrix", "projection_complement_matrix", "bottom_spd_matrix", "contraction_matrix")
        if con.args or set(con.keywords) != {name, *mapping}:
            return None
        lift = tuple(con.keywords[k] for k in mapping)
        if not all(torch.is_tensor(t) and not t.requires_grad for t in lift):
            return None
        bound = con.keywords[name]
        if torch.is_tensor(bound):
            if bound.numel() != 1:
                return None
            bound = bound.item()
        return (_lib.GABO_CONSTRAINT_MAX_EIGENVALUE_NESTED if is_max else _lib.GABO_CONSTRAINT_MIN_EIGENVALUE_NESTED, float(bound), lift)
    if con.func not in (max_eigenvalue_constraint_torch, min_eigenvalue_constraint_torch):
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


def builtin_lift(builtins):
    """The one nested mapping the nested entries of `builtins` (results of builtin_constraint) share: the (w, x0, p) device tensors of ops.nested_spd_lift_prepare for the
    kernel, None when there is no nested entry, False when they disagree (then the constraints stay host callables)."""
    lifts = [b[2] for b in builtins if b is not None and len(b) == 3]
    if not lifts:
        return None
    first = lifts[0]
    for other in lifts[1:]:
        if any(a is not b and (a.shape != b.shape or a.data_ptr() != b.data_ptr()) for a, b in zip(first, other)):
            return False
    w, v, c, k = first
    if not (5 <= w.shape[0] <= _lib.GABO_TR_NESTED_MAX_DIM):
        return False
    return ops.nested_spd_lift_prepare(w, v, c, k)
