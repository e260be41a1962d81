"""Multi-start acquisition maximisation on a manifold - same entry points and keyword arguments as the reference
(BoManifolds/manifold_optimization/manifold_optimize.py:36-321), re-organised for the MI355X:

  * the `raw_samples` candidates are scored in one batched acquisition call (as the reference does, :297-309); with
    torch.distributed initialised they are drawn and scored sharded by sample index, one all_gather of (value, sample);
  * the `num_restarts` local solves run in LOCK STEP (BatchedTrustRegions) instead of the sequential loop at :207;
  * with torch.distributed initialised, restart r is owned by rank r % world (interleaved: trust-region iteration counts
    vary), each rank optimises its share on its own GPU, and ONE all_gather of (acquisition value, candidate) per restart
    followed by a local argmax replaces get_best_candidates (:118-120).  Ties go to the lowest global restart index on
    every rank, so all ranks return the same candidate.
"""
import warnings

import numpy as np
import torch

from ..models import (BadInitialCandidatesWarning, get_best_candidates, initialize_q_batch, initialize_q_batch_nonneg,
                      is_nonnegative)
from ..fused_acquisition import FusedAcquisition
from .batched_trust_regions import BatchedProblem, BatchedTrustRegions


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _mark(options, label):
    """development hook: options["timeline"] = [] collects (label, perf_counter) pairs of the sweep's host phases (tools/sweep_native_phases.py)"""
    tl = options.get("timeline")
    if tl is not None:
        import time
        tl.append((label, time.perf_counter()))


def shard_restarts(num_restarts, rank, world):
    """Indices of the restarts rank `rank` owns (interleaved)."""
    return list(range(rank, num_restarts, world))


def gather_best(candidates_local, values_local, owned, num_restarts):
    """all_gather of per-restart (value, candidate) and argmax.  candidates_local: r_local x q x d, values_local:
    r_local.  Returns (best candidate q x d, all candidates R x q x d, all values R) - identical on every rank."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return get_best_candidates(candidates_local, values_local), candidates_local, values_local
    world = dist.get_world_size()
    per = (num_restarts + world - 1) // world
    q, d = candidates_local.shape[-2:]
    dev, dt = candidates_local.device, candidates_local.dtype
    packed = torch.full((per, 1 + q * d), float("-inf"), dtype=dt, device=dev)     # padded slots lose every argmax
    n = len(owned)
    if n:
        packed[:n, 0] = values_local.reshape(-1).to(dt)
        packed[:n, 1:] = candidates_local.reshape(n, -1)
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed)
    values = torch.full((num_restarts,), float("-inf"), dtype=dt, device=dev)
    cands = torch.zeros(num_restarts, q, d, dtype=dt, device=dev)
    for r in range(world):
        idx = shard_restarts(num_restarts, r, world)
        if idx:
            values[idx] = gathered[r][:len(idx), 0]
            cands[idx] = gathered[r][:len(idx), 1:].reshape(len(idx), q, d)
    return get_best_candidates(cands, values), cands, values


def joint_optimize_manifold(acq_function, manifold, solver, q, num_restarts, raw_samples, bounds, sample_type=torch.float64,
                            options=None, inequality_constraints=None, equality_constraints=None, pre_processing_manifold=None,
                            post_processing_manifold=None, approx_hessian=False, solver_init_conds=False):
    """Returns the `q x d` best candidate (manifold_optimize.py:36-120)."""
    options = options or {}
    analytic = getattr(acq_function, "is_analytic", True)
    _mark(options, "-> plan")
    plan = _native_sweep_plan(acq_function, manifold, solver, q if not analytic else 1, num_restarts, raw_samples, bounds, sample_type, options,
                              inequality_constraints, equality_constraints, pre_processing_manifold, post_processing_manifold, approx_hessian,
                              solver_init_conds)
    _mark(options, "<- plan")
    if plan is not None:
        return _native_sweep(plan, acq_function, solver, num_restarts, raw_samples, options)
    batch_initial_conditions = gen_batch_initial_conditions_manifold(
        acq_function=acq_function, manifold=manifold, bounds=bounds, q=None if analytic else q, num_restarts=num_restarts,
        raw_samples=raw_samples, sample_type=sample_type, options=options, post_processing_manifold=post_processing_manifold)
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    # (no broadcast: every rank selected the same initial conditions from the all_gathered raw samples)
    owned = shard_restarts(num_restarts, rank, world)
    batch_limit = options.get("batch_limit", num_restarts)
    cand_list, val_list = [], []
    start = 0
    while start < len(owned):
        idx = owned[start:start + batch_limit]
        c, v = gen_candidates_manifold(
            initial_conditions=batch_initial_conditions[idx], acquisition_function=acq_function, manifold=manifold,
            solver=solver, pre_processing_manifold=pre_processing_manifold, post_processing_manifold=post_processing_manifold,
            lower_bounds=None if bounds is None else bounds[0], upper_bounds=None if bounds is None else bounds[1],
            options={k: v for k, v in options.items() if k not in ("batch_limit", "nonnegative")},   # "hip_graphs", "device" pass through
            inequality_constraints=inequality_constraints, equality_constraints=equality_constraints,
            approx_hessian=approx_hessian, solver_init_conds=solver_init_conds)
        cand_list.append(c)
        val_list.append(v)
        start += batch_limit
    if cand_list:
        cands, vals = torch.cat(cand_list), torch.cat(val_list)
    else:
        cands = batch_initial_conditions[:0]
        vals = torch.zeros(0, dtype=batch_initial_conditions.dtype, device=batch_initial_conditions.device)
    best, _, _ = gather_best(cands, vals, owned, num_restarts)
    return best


def _library_constraint(con):
    """functools.partial over one of the library's eigenvalue constraints (plain or stated in the original space of a nested
    SPD mapping): torch code without host synchronisation, safe to capture into the hipGraphs of the trust-region iteration."""
    import functools

    from ..nested_mappings import nested_spd_constraints_utils as nested
    from ..Riemannian_utils import spd_constraints_utils_torch as plain
    return isinstance(con, functools.partial) and con.func in (
        plain.max_eigenvalue_constraint_torch, plain.min_eigenvalue_constraint_torch,
        nested.max_eigenvalue_nested_spd_constraint, nested.min_eigenvalue_nested_spd_constraint)


def gen_candidates_manifold(initial_conditions, acquisition_function, manifold, solver, pre_processing_manifold=None,
                            post_processing_manifold=None, lower_bounds=None, upper_bounds=None, inequality_constraints=None,
                            equality_constraints=None, approx_hessian=False, solver_init_conds=False, options=None):
    """Optimise every initial condition (R x 1 x d) and return (candidates R x 1 x d, acquisition values R)
    (manifold_optimize.py:124-228).  `solver` is one of this package's trust-region classes (TrustRegions, ConstrainedTrustRegions,
    StrictConstrainedTrustRegions = BatchedTrustRegions: all restarts in lock step); any other object exposing pymanopt's
    `solve(problem, x=...)` is driven restart by restart (`_gen_candidates_pointwise`)."""
    x0 = initial_conditions.detach()
    if x0.shape[1] != 1:
        raise NotImplementedError("q != 1 is not handled (neither does the reference: manifold_optimize.py:206)")
    x0 = x0[:, 0]
    if pre_processing_manifold is not None:
        x0 = pre_processing_manifold(x0)

    def cost(x):
        if post_processing_manifold is not None:
            x = post_processing_manifold(x)
        x = x[:, None].double()                     # R x (q=1) x d: a t-batch per restart  (:182-184)
        return -acquisition_function(x)

    def precon(x, d):                               # (:190-193)
        flat = d.reshape(d.shape[0], -1)
        zero = flat.sum(1) == 0
        return torch.where(zero.reshape((-1,) + (1,) * (d.dim() - 1)), d + 1e-30, d)

    from .augmented_lagrange_method import AugmentedLagrangeMethod
    if (isinstance(solver, AugmentedLagrangeMethod) and isinstance(getattr(solver, "inner_solver", None), BatchedTrustRegions)
            and not solver_init_conds and (equality_constraints is not None or inequality_constraints is not None)
            and (options or {}).get("batched_alm", True) and getattr(solver, "_logverbosity", 0) <= 0):
        # The augmented-Lagrangian method around this package's trust regions - the default solver of the reference's constrained sphere
        # examples - on all restarts in lock step: the per-restart updates of the reference's loop with the inner solves batched
        # (augmented_lagrange_method.py: solve_batched; options={"batched_alm": False} drives it restart by restart as the reference does).
        fused = FusedAcquisition.build(acquisition_function, post_processing_manifold, x0.device) if (x0.is_cuda and (options or {}).get("fused_acquisition", True)) else None
        problem = BatchedProblem(manifold, cost, approx_hessian=True, precon=precon, fused=fused)
        opt_x = solver.solve_batched(problem, x0.double(), eq_constraints=equality_constraints, ineq_constraints=inequality_constraints)
        candidates = opt_x if post_processing_manifold is None else post_processing_manifold(opt_x)
        candidates = candidates[:, None]
        with torch.no_grad():
            batch_acquisition = -problem.cost(opt_x) if fused is not None else acquisition_function(candidates)
        return candidates.detach(), batch_acquisition.detach()
    if not isinstance(solver, BatchedTrustRegions):
        return _gen_candidates_pointwise(x0, acquisition_function, manifold, solver, post_processing_manifold, inequality_constraints,
                                         equality_constraints, approx_hessian, solver_init_conds)
    fused = None
    if (options or {}).get("fused_acquisition", True) and x0.is_cuda:
        # built-in surrogate + kernel + Mandel post-processing: value and gradient as a fixed chain of HIP launches
        fused = FusedAcquisition.build(acquisition_function, post_processing_manifold, x0.device)
    problem = BatchedProblem(manifold, cost, approx_hessian=approx_hessian, precon=precon,
                             use_hip_graphs=bool((options or {}).get("hip_graphs", False)), fused=fused)
    problem.reference_precon = True          # `precon` above is the one csrc/spd_tcg.hip implements
    problem.device_tcg = bool((options or {}).get("device_tcg", True))
    problem.device_outer = bool((options or {}).get("device_outer", True))
    problem.device_iteration = bool((options or {}).get("device_iteration", True))
    problem.device_solve = bool((options or {}).get("device_solve", True))
    # the constraint callables are user code: they run eagerly between graph replays unless the caller states they are capturable -
    # or they are this library's own eigenvalue constraints with their bounds bound by functools.partial (sync-free by construction)
    cons = list(equality_constraints or []) + list(inequality_constraints or [])
    problem.capture_constraints = bool((options or {}).get("capture_constraints", bool(cons) and all(_library_constraint(c) for c in cons)))
    if solver_init_conds:
        x0 = torch.stack([torch.as_tensor(manifold.rand()) for _ in range(x0.shape[0])]).to(x0)
    if equality_constraints is not None or inequality_constraints is not None:
        opt_x = solver.solve(problem, x0.double(), eq_constraints=equality_constraints, ineq_constraints=inequality_constraints)
    else:
        opt_x = solver.solve(problem, x0.double())
    candidates = opt_x
    if post_processing_manifold is not None:
        candidates = post_processing_manifold(candidates)
    candidates = candidates[:, None]
    with torch.no_grad():
        final = getattr(solver, "log", None) or {}
        if fused is not None and final.get("one_launch_solve") and final.get("final_cost") is not None and final["final_cost"].shape[0] == opt_x.shape[0]:
            # the single-launch solve ended with the cost of every restart's final iterate in its state (evaluated by the same device
            # function the fused chain calls): no further launch
            batch_acquisition = -final["final_cost"]
        elif fused is not None:
            batch_acquisition = -fused.cost(opt_x)          # same values through the fused chain (one launch for the SPD kernels)
        else:
            batch_acquisition = acquisition_function(candidates)
    if candidates.is_cuda:
        from .. import ops as _ops
        _ops.check_deferred()          # (no-op unless a deferred launch check is still queued: then one 8-byte read-back)
    return candidates.detach(), batch_acquisition.detach()


def _gen_candidates_pointwise(x0, acquisition_function, manifold, solver, post_processing_manifold, inequality_constraints,
                              equality_constraints, approx_hessian, solver_init_conds):
    """The reference's own loop (manifold_optimize.py:175-228) for a solver that is not one of this package's lock-step trust regions:
    any object with pymanopt's `solve(problem, x=ndarray[, eq_constraints=, ineq_constraints=])`.  One pymanopt-style `Problem` on
    single points (autograd through the acquisition function, whose kernel evaluations are still the HIP kernels), restarts one
    after the other on the host."""
    import types

    from ..pymanopt_addons.problem import Problem
    from .approximate_hessian import get_hessianfd

    def cost(x):                                     # (:177-185)
        if post_processing_manifold is not None:
            x = post_processing_manifold(x)
        return -acquisition_function(x[None].double()).sum()

    def precon(x, d):                                # (:189-192)
        if np.sum(d) == 0.0:
            d += 1e-30
        return d

    problem = Problem(manifold=manifold, cost=cost, verbosity=0, arg=torch.Tensor(), precon=precon)
    if approx_hessian:
        problem._hess = types.MethodType(get_hessianfd, problem)
    constrained = equality_constraints is not None or inequality_constraints is not None
    points = []
    for i in range(x0.shape[0]):
        if solver_init_conds:
            opt_x = solver.solve(problem)
        elif constrained:
            opt_x = solver.solve(problem, x=x0[i].detach().cpu().numpy(), eq_constraints=equality_constraints,
                                 ineq_constraints=inequality_constraints)
        else:
            opt_x = solver.solve(problem, x=x0[i].detach().cpu().numpy())
        points.append(torch.as_tensor(np.asarray(opt_x), dtype=torch.float64))
    candidates = torch.stack(points).to(x0.device)
    if post_processing_manifold is not None:
        candidates = post_processing_manifold(candidates)
    candidates = candidates[:, None]
    with torch.no_grad():
        batch_acquisition = acquisition_function(candidates)
    return candidates.detach(), batch_acquisition.detach()


class _rank_stream:
    """Host draws of one rank's shard of the raw samples.  Single process: numpy's global stream as it is (the reference's draw order,
    pinned by golden vectors).  Several ranks: every rank takes ONE integer from the global stream - identically seeded ranks stay in
    step, which the GP fit's numpy candidates and every later draw rely on - and draws its shard from a temporary state seeded with
    (that integer, rank); the global state is put back afterwards.  The shards of different ranks are then distinct streams whatever the
    seeding of the processes, and nobody has to reseed numpy per rank."""

    def __init__(self, rank, world):
        self.rank, self.world, self.saved = rank, world, None

    def __enter__(self):
        if self.world > 1:
            base = int(np.random.randint(0, 2 ** 31 - 1))
            self.saved = np.random.get_state()
            np.random.seed([base, self.rank])
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            np.random.set_state(self.saved)
        return False


def _draw_raw_samples(manifold, total, first, count, options, sample_type, rank=0, world=1):
    """Raw samples first ... first + count - 1 of the `total` of one attempt, count x 1 x (point shape) (manifold_optimize.py:288).
    `manifold.rand` stays a host callable (callers monkey-patch it); two opt-in faster routes draw the same distribution."""
    device = options.get("device")
    if options.get("device_rand") and device is not None and hasattr(manifold, "rand_batch_device"):
        # drawn on the device, stream addressed by the global sample index: the shards of ranks with a common numpy seed are
        # exactly the samples one rank would have drawn
        return manifold.rand_batch_device(total, device, first=first, count=count)[:, None].to(sample_type)
    with _rank_stream(rank, world):
        if options.get("batched_rand") and hasattr(manifold, "rand_batch"):
            # one vectorised host draw (same distribution, not numpy's draw order of `count` manifold.rand() calls)
            pts = torch.as_tensor(np.asarray(manifold.rand_batch(count)))[:, None].to(sample_type)
        elif count:
            pts = torch.cat([torch.as_tensor(np.asarray(manifold.rand()))[None, None] for _ in range(count)]).to(sample_type)
        else:
            pts = torch.as_tensor(np.asarray(manifold.rand()))[None, None][:0].to(sample_type)
    return pts if device is None else pts.to(device)


def _score_raw_samples(acq_function, X_rnd, post_processing_manifold, options):
    """(post-processed samples, acquisition values) of a block of raw samples, no gradients (manifold_optimize.py:291-309)."""
    fused = None
    if options.get("fused_acquisition", True) and X_rnd.is_cuda and X_rnd.shape[1] == 1:
        fused = FusedAcquisition.build(acq_function, post_processing_manifold, X_rnd.device)
    with torch.no_grad():
        if fused is not None:          # built-in surrogate: one launch of the fused chain for the SPD kernels, same values
            Y = -fused.cost(X_rnd[:, 0].contiguous()) if X_rnd.shape[0] else X_rnd.new_zeros(0, dtype=torch.float64)
            X = X_rnd if post_processing_manifold is None else post_processing_manifold(X_rnd)
            return X, Y
        X = X_rnd if post_processing_manifold is None else post_processing_manifold(X_rnd)
        step = options.get("batch_limit") or max(X.shape[0], 1)
        parts = [acq_function(X[s:s + step]) for s in range(0, X.shape[0], step)]
        Y = torch.cat(parts).to(X) if parts else X.new_zeros(0)
    return X, Y


def _gather_raw_samples(X_loc, Y_loc, total, seed):
    """ONE all_gather that assembles the sample-index shards of every rank - rows [row_block(total, r, world)) from rank r - and
    carries each rank's proposal for the selection seed; rank 0's is the one every rank uses.  Returns (X, Y, seed), identical on
    every rank (raw-sample scoring sharded by sample index)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return X_loc, Y_loc, seed
    from ..distributed import row_block
    world = dist.get_world_size()
    per = (total + world - 1) // world
    feat = X_loc[0].numel() if X_loc.shape[0] else int(np.prod(X_loc.shape[1:]))
    packed = torch.zeros(per + 1, 1 + feat, dtype=torch.float64, device=X_loc.device)
    packed[0, 0] = float(seed)                      # < 2^52: exact in a double
    n = X_loc.shape[0]
    packed[1:1 + n, 0] = Y_loc.reshape(-1).double()
    packed[1:1 + n, 1:] = X_loc.reshape(n, -1).double()
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed)
    counts = [row_block(total, r, world) for r in range(world)]
    X = torch.cat([parts[r][1:1 + hi - lo, 1:] for r, (lo, hi) in enumerate(counts)]).reshape((total,) + tuple(X_loc.shape[1:])).to(X_loc.dtype)
    Y = torch.cat([parts[r][1:1 + hi - lo, 0] for r, (lo, hi) in enumerate(counts)]).to(Y_loc.dtype)
    if world > 1 and counts[1][1] - counts[1][0] == counts[0][1] and counts[0][1] > 0 and torch.equal(parts[0][1:], parts[1][1:]) \
            and float(parts[0][0, 0]) == float(parts[1][0, 0]) and not getattr(_gather_raw_samples, "_warned", False):
        _gather_raw_samples._warned = True
        warnings.warn("ranks 0 and 1 drew identical raw samples: manifold.rand does not read numpy's global generator (whose per-rank "
                      "streams are derived internally) - make the sampler rank-aware, or draw on the device (options['device_rand']), "
                      "whose stream is addressed by sample index.  Do NOT reseed numpy per rank: the GP fit draws from it on every rank")
    return X, Y, int(parts[0][0, 0].item())


def _selection(acq_function, options):
    """(nonneg, eta, alpha) of the botorch heuristic gen_batch_initial_conditions_manifold applies to the scored raw samples:
    initialize_q_batch_nonneg for acquisition functions that are non-negative (or options["nonnegative"]), initialize_q_batch otherwise
    (manifold_optimize.py:296-317), with their keyword arguments from `options`."""
    nonneg = bool(options.get("nonnegative") or is_nonnegative(acq_function))
    return nonneg, float(options.get("eta", 1.0)), float(options.get("alpha", 1e-4))


def select_rows(y, n, generator, nonneg, eta=1.0, alpha=1e-4):
    """Which n of the scored raw samples become restarts: models.initialize_q_batch_nonneg / initialize_q_batch ([3P] botorch.optim.initializers)
    stated on ROW INDICES, for values that are already on the host.  y: (total,) float64 numpy array.  Returns (int64 index array of length n,
    bad) - bad: the heuristic had to pick at random (botorch's BadInitialCandidatesWarning; the caller retries with more samples).
    The comparisons and index bookkeeping are numpy; everything that feeds the random draw (the weights, torch.multinomial / torch.randperm on
    `generator`) is the torch arithmetic of the functions it restates, so the rows picked are theirs, draw for draw (tests/test_selection_cpu.py).
    Rounds 4-5 called those functions on torch tensors: a dozen tiny CPU-tensor operations, 0.16-0.25 ms of a 1.4-ms sweep."""
    total = int(y.shape[0])
    if n > total:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({total})")
    if n == total:
        return np.arange(total, dtype=np.int64), False
    if not nonneg:
        Y = torch.from_numpy(y)
        Ystd = Y.std()
        if Ystd == 0:
            return torch.randperm(n=total, generator=generator)[:n].numpy().astype(np.int64), True
        max_idx = int(torch.max(Y, dim=0)[1])
        etaZ = eta * ((Y - Y.mean()) / Ystd)
        weights = torch.exp(etaZ)
        while torch.isinf(weights).any():
            etaZ *= 0.5
            weights = torch.exp(etaZ)
        idcs = torch.multinomial(weights, n, generator=generator).numpy().astype(np.int64)
        if max_idx not in idcs:
            idcs[-1] = max_idx
        return idcs, False
    max_idx = int(np.argmax(y))                 # (first maximum; a NaN wins, as in torch.max)
    max_val = float(y[max_idx])
    if max_val != max_val:
        raise RuntimeError("select_rows: the acquisition values of the raw samples contain NaN")       # (botorch's loop below would never end)
    if max_val <= 0:
        return torch.randperm(n=total, generator=generator)[:n].numpy().astype(np.int64), True
    pos = y > 0
    num_pos = int(pos.sum())
    if num_pos < n:
        remaining = n - num_pos
        rand_idx = torch.randperm(total - num_pos, generator=generator)[:remaining].numpy()
        both = np.concatenate([np.nonzero(pos)[0], np.nonzero(~pos)[0][rand_idx]])
        return both[torch.randperm(n, generator=generator).numpy()].astype(np.int64), False
    alpha_pos = y >= alpha * max_val
    while alpha_pos.sum() < n:
        alpha = 0.1 * alpha
        alpha_pos = y >= alpha * max_val
    pool = np.nonzero(alpha_pos)[0]
    weights = torch.exp(eta * (torch.from_numpy(y[alpha_pos]) / max_val - 1))
    idcs = pool[torch.multinomial(weights, n, generator=generator).numpy()].astype(np.int64)
    if max_idx not in idcs:
        idcs[-1] = max_idx
    return idcs, False


def _native_sweep_plan(acq_function, manifold, solver, q, num_restarts, raw_samples, bounds, sample_type, options, inequality_constraints,
                       equality_constraints, pre_processing_manifold, post_processing_manifold, approx_hessian, solver_init_conds):
    """The sweep as two native host calls (csrc/spd_sweep.hip: gabo_spd_sweep_score / gabo_spd_sweep_solve) when EVERY launch of it would be
    one the native driver issues - built-in SPD surrogate evaluated in one launch, raw samples drawn on the device, eigenvalue bounds built
    with functools.partial (or no constraints), FD Hessian, the whole solve one launch, one process.  Returns what the driver needs, or None:
    then the Python path below runs, launch for launch the same work.  options={"native_sweep": False} keeps the Python path."""
    from ..manifolds import PositiveDefinite, Sphere
    from ..Riemannian_utils import spd_utils_torch
    from ..Riemannian_utils.spd_constraints_utils_torch import builtin_constraint
    from .. import _lib
    device = options.get("device")
    if not (options.get("native_sweep", True) and device is not None):
        return None
    if isinstance(manifold, Sphere):
        if _dist() is not None:
            return None       # (the sphere twin has no sharded form: the Python path below shards it)
        # the sphere twin (gabo_sphere_sweep_score / _solve): stock trust regions without constraints, exact or FD Hessian, host sampler
        if not (q == 1 and bounds is None and not solver_init_conds and sample_type == torch.float64 and isinstance(solver, BatchedTrustRegions)
                and not solver.use_rand and solver.maxtime >= 1000 and solver.trace is None and not equality_constraints
                and not inequality_constraints and pre_processing_manifold is None and post_processing_manifold is None):
            return None
        if any(options.get(k, True) is False for k in ("fused_acquisition", "device_tcg", "device_outer", "device_iteration", "device_solve")):
            return None
        if options.get("batch_limit", num_restarts) < num_restarts or num_restarts < 1 or raw_samples < 1:
            return None
        dev = torch.device(device)
        if dev.type != "cuda":
            return None
        fused = FusedAcquisition.build(acq_function, None, dev)
        if fused is None or not (fused.family == "sphere" and fused.single_launch):
            return None
        return {"fused": fused, "device": dev, "manifold": manifold, "builtins": [], "device_rand": False, "sphere": True,
                "exact_hessian": not approx_hessian}
    if not (q == 1 and bounds is None and not solver_init_conds and approx_hessian and sample_type == torch.float64
            and isinstance(solver, BatchedTrustRegions) and not solver.use_rand and solver.maxtime >= 1000
            and solver.trace is None and not equality_constraints):
        return None
    if any(options.get(k, True) is False for k in ("fused_acquisition", "device_tcg", "device_outer", "device_iteration", "device_solve")):
        return None
    if options.get("batch_limit", num_restarts) < num_restarts or num_restarts < 1 or raw_samples < 1:
        return None
    if not (isinstance(manifold, PositiveDefinite) and 2 <= manifold._n <= 8):
        return None
    device_rand = bool(options.get("device_rand")) and hasattr(manifold, "rand_batch_device")
    if device_rand and not (type(manifold).rand_batch_device is PositiveDefinite.rand_batch_device and "rand_batch_device" not in vars(manifold)
                            and hasattr(manifold, "min_eig") and hasattr(manifold, "max_eig")):
        return None       # (a sampler of the caller's own on the device: the Python path calls it)
    if pre_processing_manifold is not spd_utils_torch.vector_to_symmetric_matrix_mandel_torch:
        return None
    cons = list(inequality_constraints or [])
    builtins = [builtin_constraint(c) for c in cons]
    if len(cons) > 8 or any(b is None or len(b) != 2 or b[0] not in (_lib.GABO_CONSTRAINT_MAX_EIGENVALUE, _lib.GABO_CONSTRAINT_MIN_EIGENVALUE)
                            for b in builtins):
        return None
    dev = torch.device(device)
    if dev.type != "cuda":
        return None
    fused = FusedAcquisition.build(acq_function, post_processing_manifold, dev)
    if fused is None or not (fused.family == "spd" and fused.flavour in ("ai", "le") and fused.single_launch and fused.matrix_input):
        return None       # (affine-invariant and log-Euclidean surrogates: the metrics gabo_spd_tr_solve iterates)
    import ctypes
    if not _lib.load().gabo_spd_tr_solve_supported(ctypes.byref(fused.acq_params()), int(num_restarts), int(manifold._n), len(builtins), 0):
        return None       # (e.g. the log-Euclidean surrogate at d = 7, 8 beyond its LDS-resident form: the propose / update launches run)
    return {"fused": fused, "device": dev, "manifold": manifold, "builtins": builtins, "device_rand": device_rand}


_sweep_workspaces = {}


def _all_gather_rows(dist, full, block):
    """full (world * rows, width) <- every rank's block (rows, width): ONE collective; backends without all_gather_into_tensor take the list form"""
    try:
        dist.all_gather_into_tensor(full, block)
    except (RuntimeError, NotImplementedError, AttributeError):
        parts = [torch.empty_like(block) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, block)
        full.copy_(torch.cat(parts))


class _RowsState:
    """What one (device, stream) keeps between sweeps: the device workspace of the native driver (its two tables as tensors) and a block of PAGE-LOCKED
    host memory the kernels write straight into (scores, result rows) and read from (picked rows) - no copy launch, no pageable staging; the host
    looks at it through numpy once the stream has drained.  Keyed by what fixes the tables (d, rows, restarts, constraints); the number of training
    points only sizes the solver's scratch behind them, so a BO loop whose training set grows by one point per iteration keeps its state and lets the
    workspace grow when it has to."""

    def __init__(self, lib, dev, n_train, d, max_rows, restarts, n_constraints):
        self.key = (d, max_rows, restarts, n_constraints)
        self.dev = dev
        dv = d * (d + 1) // 2
        self.ws, self.wsb = None, 0
        self.fit(lib, n_train)
        self.pinned = torch.zeros(3 + max_rows + restarts + restarts * (2 + dv), dtype=torch.float64).pin_memory()
        host = self.pinned.numpy()
        self.err = host[:3].view(np.int32)                  # mirrors of (score status, solve status, selection flag): int32[2] each
        self.values = host[3:3 + max_rows]
        self.picked = host[3 + max_rows:3 + max_rows + restarts].view(np.int64)
        self.results = host[3 + max_rows + restarts:].reshape(restarts, 2 + dv)
        base = self.pinned.data_ptr()
        self.err_ptr = (base, base + 8, base + 16)
        self.values_ptr, self.picked_ptr, self.results_ptr = base + 24, base + 8 * (3 + max_rows), base + 8 * (3 + max_rows + restarts)
        self.status = torch.zeros(3, 2, dtype=torch.int32, device=dev)      # the device words behind the mirrors (only a failing launch writes them)
        self.status_ptr = tuple(self.status.data_ptr() + 8 * k for k in range(3))
        self.picked_dev = torch.zeros(restarts, dtype=torch.int64, device=dev)      # device selection: this rank's rows
        self.samples_dev = None                                                     # ... and every restart's sample index (sized on first use)

    def fit(self, lib, n_train):
        """the workspace for a surrogate on n_train points: grown when the one held is too small (the tables sit at its start, at offsets that do not
        depend on n_train)"""
        import ctypes
        d, max_rows, restarts, n_constraints = self.key
        dv = d * (d + 1) // 2
        need = int(lib.gabo_spd_sweep_rows_workspace_bytes(n_train, d, max_rows, restarts, n_constraints))
        if self.ws is not None and need <= self.wsb:
            return
        self.wsb = need + need // 4           # (room for a few more training points before the next growth)
        self.ws = torch.empty(self.wsb // 8 + 1, dtype=torch.float64, device=self.dev)
        raw_p, res_p = ctypes.c_void_p(), ctypes.c_void_p()
        if lib.gabo_spd_sweep_rows_tables(self.ws.data_ptr(), n_train, d, max_rows, restarts, n_constraints, ctypes.byref(raw_p), ctypes.byref(res_p)) != 0:
            raise RuntimeError("gabo_spd_sweep_rows_tables refused the workspace")
        o_raw, o_res = (raw_p.value - self.ws.data_ptr()) // 8, (res_p.value - self.ws.data_ptr()) // 8
        self.raw = self.ws[o_raw:o_raw + max_rows * (1 + dv)].view(max_rows, 1 + dv)
        self.res = self.ws[o_res:o_res + restarts * (2 + dv)].view(restarts, 2 + dv)

    def raise_if_failed(self, which, what):
        """after the stream has drained: did a launch of call `which` (0 score, 1 solve) report a non-SPD matrix?  Host memory only."""
        e = self.err[2 * which:2 * which + 2]
        if e[0] != 0:
            idx = int(e[1])
            e[:] = 0
            self.status[which].zero_()
            raise RuntimeError(f"{what}: input matrix #{idx} is not positive definite (Cholesky pivot <= 0)")


def _rows_state(lib, dev, stream, n_train, d, max_rows, restarts, n_constraints):
    key = (dev.index, stream)
    st = _sweep_workspaces.get(key)
    if st is None or st.key != (d, max_rows, restarts, n_constraints):
        st = _sweep_workspaces[key] = _RowsState(lib, dev, n_train, d, max_rows, restarts, n_constraints)
    else:
        st.fit(lib, n_train)
    return st


def _native_sweep(plan, acq_function, solver, num_restarts, raw_samples, options):
    """joint_optimize_manifold through the native driver (csrc/spd_sweep.hip): gabo_spd_sweep_score_rows -> selection on the host ->
    gabo_spd_sweep_solve_rows -> argmin - five launches per sweep on one GPU; with the selection on the device (the default) one host wait.  The draws from numpy's and torch's generators, the
    selection heuristic and every statement executed on the device are those of the Python path, so the candidate returned is the same, bit for bit
    (tests/test_gpu_native_sweep.py).

    With torch.distributed initialised: rank r draws and scores raw samples [r * per, (r + 1) * per) into its block of the raw-row table,
    ONE all_gather assembles the table (each block led by a header row that carries the rank's proposal for the selection seed; rank 0's is used),
    every rank selects the same rows, solves restarts r, r + P, r + 2P, ... and ONE all_gather of the result rows followed by an argmax in restart
    order (first index on ties) replaces get_best_candidates (manifold_optimize.py:118-120).  Exactly two collectives, none inside the driver."""
    import ctypes
    import time

    from .. import _lib, ops
    if plan.get("sphere"):
        return _native_sweep_sphere(plan, acq_function, solver, num_restarts, raw_samples, options)
    lib = _lib.load()
    fused, dev, man = plan["fused"], plan["device"], plan["manifold"]
    d, dv, R = man._n, man._n * (man._n + 1) // 2, int(num_restarts)
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    cfg = _lib.SweepConfig()
    cfg.acq = fused.acq_params()
    cfg.d = d
    cfg.min_eig, cfg.max_eig = (float(man.min_eig), float(man.max_eig)) if plan["device_rand"] else (0.0, 0.0)
    cfg.n_constraints = len(plan["builtins"])
    for k, b in enumerate(plan["builtins"]):
        cfg.constraint_kind[k], cfg.constraint_bound[k] = int(b[0]), float(b[1])
    cfg.strict = 1 if solver.strict_constraints else 0
    delta_bar = getattr(man, "typicaldist", None) or float(man.dim) ** 0.5          # (BatchedTrustRegions._solve's defaults)
    cfg.delta_bar, cfg.delta0, cfg.delta_cons = float(delta_bar), float(delta_bar) / 8, 1e-6
    cfg.theta, cfg.kappa, cfg.mininner, cfg.maxinner = float(solver.theta), float(solver.kappa), 1, int(man.dim)
    cfg.rho_prime, cfg.rho_regularization = float(solver.rho_prime), float(solver.rho_regularization)
    cfg.mingradnorm, cfg.maxiter = float(solver.mingradnorm), int(solver.maxiter)
    nonneg, eta, alpha = _selection(acq_function, options)
    cfg_ref = ctypes.byref(cfg)
    checking = ops._check_errors is not False
    # the selection on the device (gabo_spd_sweep_select_rows) when it is botorch's non-negative heuristic on a table the kernel takes
    on_device = bool(options.get("device_selection", True)) and nonneg and bool(lib.gabo_spd_sweep_select_supported(raw_samples, R))
    r_loc, r_per = len(range(rank, R, world)), (R + world - 1) // world
    n_train = int(fused.train.shape[0])
    time0 = time.time()
    with torch.cuda.device(dev):
        stream = ops._stream_ptr(dev)
        picked = None
        for attempt in range(1, 5):                     # the reference's factor = 1 ... max_factor - 1 (manifold_optimize.py:283-320)
            total = raw_samples * attempt
            per = (total + world - 1) // world          # samples per rank; a rank's block of the table = one header row + per sample rows
            lo = min(rank * per, total)
            cnt = min(lo + per, total) - lo
            st = _rows_state(lib, dev, stream, n_train, d, world * (per + 1), max(r_per, 1), cfg.n_constraints)
            row0 = rank * (per + 1) + 1
            seed, raw = 0, None
            with _rank_stream(rank, world if not plan["device_rand"] else 1):
                if plan["device_rand"]:
                    seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))       # (the draw of manifolds.PositiveDefinite.rand_batch_device)
                elif options.get("batched_rand") and hasattr(man, "rand_batch"):       # the host samplers of _draw_raw_samples, same draws
                    raw = np.ascontiguousarray(man.rand_batch(cnt), dtype=np.float64)
                else:
                    raw = np.ascontiguousarray(np.stack([np.asarray(man.rand()) for _ in range(cnt)]), dtype=np.float64) if cnt else np.zeros((0, d, d))
            if raw is not None and raw.shape != (cnt, d, d):
                raise RuntimeError(f"manifold.rand returned points of shape {raw.shape[1:]}, expected ({d}, {d})")
            _mark(options, "-> score")
            pending = ops.prefetch_deferred()           # (the GP's set-up launches: their status words travel while the scoring launches run)
            if cnt:
                rc = lib.gabo_spd_sweep_score_rows(cfg_ref, lo, row0, cnt, world * (per + 1), max(r_per, 1), seed & 0xFFFFFFFFFFFFFFFF,
                                                   None if raw is None else raw.ctypes.data, st.values_ptr if (world == 1 and not on_device) else None,
                                                   st.ws.data_ptr(), st.wsb, st.status_ptr[0], st.err_ptr[0], 1 if (world == 1 and not on_device) else 0,
                                                   stream)
                if rc != 0:
                    _lib.check(rc, "gabo_spd_sweep_score_rows")
            _mark(options, "<- score")
            sel_seed = int(torch.randint(0, 2 ** 52, (1,)).item())
            if on_device:
                # score -> (all_gather) -> selection kernel -> solve, back to back on the stream: the host waits once, at the end
                if world > 1:
                    block = st.raw[rank * (per + 1):(rank + 1) * (per + 1)]
                    block[0, 0] = float(sel_seed)
                    _all_gather_rows(dist, st.raw, block.clone())
                if st.samples_dev is None or st.samples_dev.numel() < R:
                    st.samples_dev = torch.zeros(R, dtype=torch.int64, device=dev)
                st.err[4] = -1
                rc = lib.gabo_spd_sweep_select_rows(st.raw.data_ptr(), d, total, per, R, eta, alpha, sel_seed, 1 if world > 1 else 0, rank, world,
                                                    st.picked_dev.data_ptr(), st.samples_dev.data_ptr(), st.status_ptr[2], st.err_ptr[2], stream)
                if rc != 0:
                    _lib.check(rc, "gabo_spd_sweep_select_rows")
                break
            if world == 1:
                y = st.values[:total]
            else:
                # the ONE collective of this stage: every rank's block (header + rows) -> the whole table, in place on every rank
                block = st.raw[rank * (per + 1):(rank + 1) * (per + 1)]
                block[0, 0] = float(sel_seed)           # < 2^52: exact in a double
                _all_gather_rows(dist, st.raw, block.clone())
                col = st.raw[:, 0].cpu().numpy().reshape(world, per + 1)         # (this copy waits for the stream)
                sel_seed = int(col[0, 0])
                y = np.ascontiguousarray(col[:, 1:].reshape(-1)[:total])
            ops.check_prefetched(pending)
            if checking:
                st.raise_if_failed(0, "gabo_spd_sweep_score_rows")
            _mark(options, "<- checks")
            gen = torch.Generator()
            gen.manual_seed(sel_seed)
            picked, bad = select_rows(y, R, gen, nonneg, eta, alpha)
            if not bad:
                break
        else:
            warnings.warn("Unable to find non-zero acquisition function values - initial conditions are being selected randomly.",
                          BadInitialCandidatesWarning)
        _mark(options, "<- selection")
        if not on_device:
            mine = picked[rank::world]                   # restart k belongs to rank k % world (trust-region iteration counts vary: interleaved)
            st.picked[:r_loc] = (mine // per) * (per + 1) + 1 + mine % per          # sample index -> row of the table
        if world > 1 and r_loc < r_per:
            st.res[r_loc:, 0] = float("inf")             # (a rank with one restart fewer: its padding row loses every argmax)
        if r_loc:
            rc = lib.gabo_spd_sweep_solve_rows(cfg_ref, st.picked_dev.data_ptr() if on_device else st.picked_ptr, r_loc, world * (per + 1),
                                               st.results_ptr if world == 1 else None, st.status_ptr[2] if on_device else None, st.ws.data_ptr(), st.wsb,
                                               st.status_ptr[1], st.err_ptr[1],
                                               1 if world == 1 else 0, stream)
            if rc != 0:
                _lib.check(rc, "gabo_spd_sweep_solve_rows")
        _mark(options, "<- solve")
        if world == 1:
            rows = st.results[:R]
        else:
            gathered = torch.empty(world * r_per, 2 + dv, dtype=torch.float64, device=dev)
            _all_gather_rows(dist, gathered, st.res[:r_per].clone())
            # rank r's row j is restart j * world + r           (the copy to the host waits for the stream)
            rows = gathered.cpu().numpy().reshape(world, r_per, 2 + dv).transpose(1, 0, 2).reshape(world * r_per, 2 + dv)[:R]
        if on_device:
            ops.check_prefetched(pending)
            if checking:
                st.raise_if_failed(0, "gabo_spd_sweep_score_rows")
            if st.err[4] != 0:
                # the heuristic needs its random fall-backs (no positive value among the raw samples, or fewer than restarts), a value is NaN - or the
                # flag never arrived: the host heuristic decides, with its retries (manifold_optimize.py:283-320)
                # (the launches behind the selection did no work: gabo_spd_sweep_solve_rows' skip_flag)
                return _native_sweep(plan, acq_function, solver, num_restarts, raw_samples, dict(options, device_selection=False))
        if checking:
            st.raise_if_failed(1, "gabo_spd_sweep_solve_rows")
    cost = rows[:, 0]
    best = int(np.argmin(cost))          # torch.argmax(-cost): the first of equal values; a NaN wins (numpy's argmin returns the first NaN too)
    solver.log = {"iterations": int(rows[:, 1].max()), "per_restart_iterations": torch.from_numpy(rows[:, 1].astype(np.int64)),
                  "final_cost": torch.from_numpy(cost.copy()), "final_gradnorm": None, "cost_evals": 0, "grad_evals": 0,
                  "time": time.time() - time0, "one_launch_solve": True, "native_sweep": True, "world_size": world,
                  "device_selection": on_device}
    if on_device and options.get("log_picked"):
        solver.log["picked_samples"] = st.samples_dev[:R].cpu().numpy()
    if world == 1:
        out = st.res[best, 2:].clone().reshape(1, dv)          # (the winner's row of the device table: no host -> device copy)
    else:
        out = torch.from_numpy(rows[best, 2:].copy()).reshape(1, dv).to(dev)
    _mark(options, "<- result")
    return out


def _native_sweep_sphere(plan, acq_function, solver, num_restarts, raw_samples, options):
    """the sphere twin of _native_sweep: gabo_sphere_sweep_score / gabo_sphere_sweep_solve around the same selection heuristic and host sampler"""
    import ctypes
    import time

    from .. import _lib, ops
    lib = _lib.load()
    fused, dev, man = plan["fused"], plan["device"], plan["manifold"]
    dim, R = int(man._n), int(num_restarts)
    cfg = _lib.SphereSweepConfig()
    cfg.acq = fused.sphere_acq_params()
    delta_bar = getattr(man, "typicaldist", None) or float(man.dim) ** 0.5
    cfg.delta_bar, cfg.delta0 = float(delta_bar), float(delta_bar) / 8
    cfg.theta, cfg.kappa, cfg.mininner, cfg.maxinner = float(solver.theta), float(solver.kappa), 1, int(man.dim)
    cfg.exact_hessian = 1 if plan["exact_hessian"] else 0
    cfg.rho_prime, cfg.rho_regularization = float(solver.rho_prime), float(solver.rho_regularization)
    cfg.mingradnorm, cfg.maxiter = float(solver.mingradnorm), int(solver.maxiter)
    nonneg, eta, alpha = _selection(acq_function, options)
    time0 = time.time()

    def draw(total):          # the host samplers of _draw_raw_samples, same draws
        if options.get("batched_rand") and hasattr(man, "rand_batch"):
            raw = np.ascontiguousarray(man.rand_batch(total), dtype=np.float64)
        else:
            raw = np.ascontiguousarray(np.stack([np.asarray(man.rand()) for _ in range(total)]), dtype=np.float64)
        if raw.shape != (total, dim):
            raise RuntimeError(f"manifold.rand returned points of shape {raw.shape[1:]}, expected ({dim},)")
        return raw

    def workspace(total):
        wsb = int(lib.gabo_sphere_sweep_workspace_bytes(dim, total, R))
        key = ("sphere", dev.index, stream)
        ws = _sweep_workspaces.get(key)
        if ws is None or ws.numel() < wsb:
            ws = _sweep_workspaces[key] = torch.empty(wsb, dtype=torch.uint8, device=dev)
        return ws, wsb

    def result(ws, best, value, iters, cand_p, cost_p, it_p, picked_p=None, device_selection=False):
        base = ws.data_ptr()

        def view(ptr, count, dtype):
            off = int(ptr.value) - base
            return ws[off:off + 8 * count].view(dtype)
        cands = view(cand_p, R * dim, torch.float64).reshape(R, dim)
        solver.log = {"iterations": int(iters.value), "per_restart_iterations": view(it_p, R, torch.int64).clone(),
                      "final_cost": view(cost_p, R, torch.float64).clone(), "final_gradnorm": None, "cost_evals": 0, "grad_evals": 0,
                      "time": time.time() - time0, "one_launch_solve": True, "native_sweep": True, "device_selection": device_selection}
        if picked_p is not None and options.get("log_picked"):
            solver.log["picked"] = view(picked_p, R, torch.int64).cpu().numpy().copy()
        return cands[int(best.value)].reshape(1, dim).clone()

    with torch.cuda.device(dev):
        stream = ops._stream_ptr(dev)
        # ONE call, one wait (gabo_sphere_sweep_run): the selection heuristic as a kernel between the scoring and the solve (the SPD sweep's), the raw
        # samples from the host sampler or - options["device_rand"] - drawn on the device.  Not for acquisition functions that can be negative, more raw
        # samples than the kernel holds, or when it reports that the heuristic needs its random fall-backs: then the two-call path below runs.
        if (options.get("device_selection", True) and nonneg and lib.gabo_spd_sweep_select_supported(raw_samples, R)):
            total = raw_samples
            ws, wsb = workspace(total)
            raw = None if options.get("device_rand") else draw(total)
            sample_seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64)) if raw is None else 0
            sel_seed = int(torch.randint(0, 2 ** 52, (1,)).item())
            best, iters, fallback = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
            value = ctypes.c_double(0.0)
            cand_p, cost_p, it_p, picked_p = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(lib.gabo_sphere_sweep_run(ctypes.byref(cfg), total, R, None if raw is None else raw.ctypes.data, sample_seed, float(eta), float(alpha),
                                                 sel_seed, ctypes.byref(best), ctypes.byref(value), ctypes.byref(iters), ctypes.byref(cand_p),
                                                 ctypes.byref(cost_p), ctypes.byref(it_p), ctypes.byref(picked_p), ctypes.byref(fallback), ws.data_ptr(), wsb,
                                                 stream), "gabo_sphere_sweep_run")
            ops.check_deferred()
            if not fallback.value:
                return result(ws, best, value, iters, cand_p, cost_p, it_p, picked_p, True)
        picked = None
        for attempt in range(1, 5):
            total = raw_samples * attempt
            ws, wsb = workspace(total)
            raw = draw(total)
            y = np.empty(total, dtype=np.float64)
            _lib.check(lib.gabo_sphere_sweep_score(ctypes.byref(cfg), total, total, R, raw.ctypes.data, y.ctypes.data, ws.data_ptr(), wsb, stream),
                       "gabo_sphere_sweep_score")
            ops.check_deferred()
            sel_seed = int(torch.randint(0, 2 ** 52, (1,)).item())
            gen = torch.Generator()
            gen.manual_seed(sel_seed)
            picked, bad = select_rows(y, R, gen, nonneg, eta, alpha)
            if not bad:
                break
        else:
            warnings.warn("Unable to find non-zero acquisition function values - initial conditions are being selected randomly.",
                          BadInitialCandidatesWarning)
        idx = np.ascontiguousarray(picked, dtype=np.int64)
        best, iters = ctypes.c_int64(0), ctypes.c_int64(0)
        value = ctypes.c_double(0.0)
        cand_p, cost_p, it_p = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.gabo_sphere_sweep_solve(ctypes.byref(cfg), idx.ctypes.data, R, total, ctypes.byref(best), ctypes.byref(value), ctypes.byref(iters),
                                               ctypes.byref(cand_p), ctypes.byref(cost_p), ctypes.byref(it_p), ws.data_ptr(), wsb, stream),
                   "gabo_sphere_sweep_solve")
    if options.get("log_picked"):
        out = result(ws, best, value, iters, cand_p, cost_p, it_p)
        solver.log["picked"] = idx.copy()
        return out
    return result(ws, best, value, iters, cand_p, cost_p, it_p)


def gen_batch_initial_conditions_manifold(acq_function, manifold, bounds, q, num_restarts, raw_samples,
                                          sample_type=torch.float64, options=None, post_processing_manifold=None):
    """`num_restarts x q x d` initial conditions chosen among `raw_samples` random manifold points by the botorch
    heuristics (manifold_optimize.py:232-321): up to four attempts with raw_samples, 2 raw_samples, ... as long as the heuristic
    reports that it had to pick at random.

    With torch.distributed initialised the raw samples are sharded BY SAMPLE INDEX: rank r draws and scores rows
    row_block(total, r, world) only, one all_gather assembles (Y, X), and every rank runs the selection on identical data with an
    identical random stream - so all ranks hold the same initial conditions without a broadcast."""
    options = options or {}
    q = 1 if q is None else q
    nonneg, eta, alpha = _selection(acq_function, options)
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    chosen = None
    for attempt in range(1, 5):                     # the reference's factor = 1 ... max_factor - 1
        total = raw_samples * attempt * q
        lo, hi = (0, total)
        if world > 1:
            from ..distributed import row_block
            lo, hi = row_block(total, rank, world)
        X_loc, Y_loc = _score_raw_samples(acq_function, _draw_raw_samples(manifold, total, lo, hi - lo, options, sample_type, rank, world),
                                          post_processing_manifold, options)
        seed = int(torch.randint(0, 2 ** 52, (1,)).item())      # from torch's global generator, like botorch's own multinomial draw
        X_rnd, Y_rnd, seed = _gather_raw_samples(X_loc, Y_loc, total, seed)
        # The selection heuristic is a dozen data-dependent decisions on `total` scalars: on the device every one of them is a launch and a
        # read-back (0.85 ms of the 4.4-ms config-4 sweep, tools/sweep_phases2.py); here the values come to the host in ONE copy, the heuristic
        # selects ROW INDICES there (select_rows; host generator with the common seed: identical on every rank) and the rows are gathered
        # on the device.
        gen = torch.Generator()
        gen.manual_seed(seed)
        y_host = Y_rnd.detach().double().cpu().numpy()          # (the host waits for the scores here anyway: deferred launch checks cost nothing now)
        if Y_rnd.is_cuda:
            from .. import ops as _ops
            _ops.check_deferred()
        picked, bad = select_rows(np.ascontiguousarray(y_host.reshape(-1)), num_restarts, gen, nonneg, eta, alpha)
        chosen = X_rnd.index_select(0, torch.from_numpy(picked).to(X_rnd.device))
        if not bad:
            return chosen
    warnings.warn("Unable to find non-zero acquisition function values - initial conditions are being selected randomly.",
                  BadInitialCandidatesWarning)
    return chosen
