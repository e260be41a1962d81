"""The config-4 sweep through the native driver (gabo_spd_sweep_score_rows / _select_rows / _solve_rows, csrc/spd_sweep.hip) against the Python path
of joint_optimize_manifold (manifold_optimize.py:36-120 of the reference: initial conditions, the restarts' solves, argmax): the device executes the
same statements in the same order and, with the selection heuristic left on the host (device_selection=False), both random generators are consumed
as on the Python path - so the returned candidate is the same BIT FOR BIT.  With the selection on the device (the default) the picks come from the
library's own random stream; given those picks the Python path returns the same candidate, again bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _both(device_rand=True, **kw):
    from tools.sweep_bench import run_sweep
    _, best_n, val_n, log_n = run_sweep("cuda:0", device_rand=device_rand, builtin_constraint=True, native_sweep=True, device_selection=False, **kw)
    _, best_p, val_p, log_p = run_sweep("cuda:0", device_rand=device_rand, builtin_constraint=True, native_sweep=False, **kw)
    assert log_n.get("native_sweep") and not log_p.get("native_sweep")
    return (best_n, val_n, log_n), (best_p, val_p, log_p)


@pytest.mark.parametrize("restarts,raw", [(512, 2048), (64, 256), (7, 50), (1, 3)])
def test_native_sweep_returns_the_python_path_candidate_bit_for_bit(restarts, raw):
    (bn, vn, ln), (bp, vp, lp) = _both(num_restarts=restarts, raw_samples=raw)
    assert torch.equal(bn, bp) and vn == vp
    assert ln["iterations"] == lp["iterations"]
    assert torch.equal(ln["per_restart_iterations"].cpu(), lp["per_restart_iterations"].cpu())
    np.testing.assert_array_equal(ln["final_cost"].cpu().numpy(), lp["final_cost"].cpu().numpy())


def test_native_sweep_without_constraints_and_with_the_strict_solver():
    (bn, vn, _), (bp, vp, _) = _both(num_restarts=96, raw_samples=384, constraint=False)
    assert torch.equal(bn, bp) and vn == vp
    (bn, vn, _), (bp, vp, _) = _both(num_restarts=96, raw_samples=384, strict=True, maxiter=30)
    assert torch.equal(bn, bp) and vn == vp


@pytest.mark.parametrize("batched", [True, False])
def test_native_sweep_with_raw_samples_drawn_by_the_callers_host_sampler(batched):
    """`manifold.rand` is user code (the reference binds spd_sample to it, examples/gabo_spd.py:102): its draws - one by one from numpy's global
    stream, or the vectorised rand_batch - are handed to the native driver as they are"""
    (bn, vn, ln), (bp, vp, lp) = _both(device_rand=False, batched_rand=batched, num_restarts=24, raw_samples=96, maxiter=25)
    assert torch.equal(bn, bp) and vn == vp
    np.testing.assert_array_equal(ln["final_cost"].cpu().numpy(), lp["final_cost"].cpu().numpy())


def test_native_sweep_declines_what_it_does_not_cover():
    """an opaque constraint callable, a plan without the one-launch solve: the Python path runs (and says so in the solver's log)"""
    from tools.sweep_bench import run_sweep
    for kw in (dict(device_rand=True, builtin_constraint=False), dict(device_rand=True, builtin_constraint=True, device_solve=False)):
        log = run_sweep("cuda:0", num_restarts=32, raw_samples=128, maxiter=10, **kw)[3]
        assert not log.get("native_sweep")


@pytest.mark.parametrize("kw", [dict(), dict(strict=True), dict(constraint=False), dict(maxiter=250)])
def test_one_launch_solve_shortcuts_do_not_change_a_bit(kw, monkeypatch):
    """The single-launch solve evaluates a proposal's value before its gradient after a rejection and reuses the previous proposal when tCG returns
    the same step again (csrc/spd_tr_body.hpp).  With GABO_TR_NO_SHORTCUTS in the environment the same kernel runs every iteration in full (the form of
    rounds 1-4): final iterates, costs and iteration counts of all 512 restarts must agree bit for bit - including the restarts that sit on the
    eigenvalue bound and have every proposal but one rejected (radius down to 1e-150 at 250 iterations)."""
    from tools.sweep_bench import run_sweep
    monkeypatch.delenv("GABO_TR_NO_SHORTCUTS", raising=False)
    _, b1, v1, l1 = run_sweep("cuda:0", device_rand=True, builtin_constraint=True, native_sweep=False, **kw)
    monkeypatch.setenv("GABO_TR_NO_SHORTCUTS", "1")
    _, b2, v2, l2 = run_sweep("cuda:0", device_rand=True, builtin_constraint=True, native_sweep=False, **kw)
    monkeypatch.delenv("GABO_TR_NO_SHORTCUTS")
    assert l1.get("one_launch_solve") and l2.get("one_launch_solve")
    assert torch.equal(l1["per_restart_iterations"].cpu(), l2["per_restart_iterations"].cpu())
    np.testing.assert_array_equal(l1["final_cost"].cpu().numpy(), l2["final_cost"].cpu().numpy())
    assert torch.equal(b1, b2) and v1 == v2
    if kw.get("constraint", True) and not kw.get("strict"):
        assert int(l1["per_restart_iterations"].max()) == kw.get("maxiter", 100)       # (the restarts on the bound are in the set)


def test_native_sweep_on_the_log_euclidean_surrogate_of_config_5():
    """config 5's latent sweep: S^2_++, SpdLogEuclideanGaussianKernel, StrictConstrainedTrustRegions semantics, an eigenvalue box built with
    functools.partial (hd_gabo_spd.py:244-257 without the nested lift): native driver against the Python path, bit for bit"""
    import functools

    from gabotorch_amd import manifolds, models
    from gabotorch_amd.kernel_utils.kernels_spd import SpdLogEuclideanGaussianKernel
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
    from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat
    from oracle import spd as ospd
    rng = np.random.default_rng(11)
    q = np.linalg.qr(rng.standard_normal((30, 2, 2)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.1, 4.0, (30, 2)), q)
    z = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(0.5 * (X + X.transpose(0, 2, 1))), device="cuda:0")
    f = torch.tensor((np.log(np.linalg.eigvalsh(X) / 2.0) ** 2).sum(1), device="cuda:0")
    outs = []
    for native in (True, False):
        kern = SpdLogEuclideanGaussianKernel().double()
        kern.lengthscale = torch.tensor(1.5, dtype=torch.float64)
        gp = models.ExactGP(z, f, kern, outputscale=1.0, noise=1e-2)
        acq = models.ExpectedImprovement(gp, best_f=float(f.min()), maximize=False)
        man = manifolds.PositiveDefinite(2)
        man.min_eig, man.max_eig = 0.05, 5.0
        cons = [functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=5.0),
                functools.partial(scut.min_eigenvalue_constraint_torch, minimum_eigenvalue=0.05)]
        np.random.seed(7)
        torch.manual_seed(7)
        solver = BatchedTrustRegions(mingradnorm=2e-4, maxiter=60, strict_constraints=True)
        best = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=48, raw_samples=192, bounds=None,
                                       options={"device": "cuda:0", "batched_rand": True, "native_sweep": native, "device_selection": False}, inequality_constraints=cons,
                                       pre_processing_manifold=to_mat, post_processing_manifold=to_vec, approx_hessian=True)
        assert bool(solver.log.get("native_sweep")) == native and solver.log.get("one_launch_solve")
        outs.append((best.clone(), solver.log["final_cost"].clone(), solver.log["per_restart_iterations"].clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1].cpu(), outs[1][1].cpu()) and torch.equal(outs[0][2].cpu(), outs[1][2].cpu())


@pytest.mark.parametrize("approx", [False, True])
def test_native_sphere_sweep_returns_the_python_path_candidate(approx):
    """gabo_sphere_sweep_score / gabo_sphere_sweep_solve (the reference's gabo_sphere setting: stock trust regions, no constraints, exact or FD
    Hessian-vector products) against the Python path: same candidate, costs and iteration counts.  (The initial gradient NORM is a torch
    reduction in one path and a plain loop in the other; it only enters the stopping test, so the comparison stays bitwise.)"""
    from tools.sphere_sweep_bench import run
    outs = []
    for native in (True, False):
        # (device_selection=False: the heuristic on the host for both - the selection kernel draws from the library's own stream, see the next test)
        dt, val, its, log = run(approx=approx, constrained=False, device="cuda:0", native=native, R=96, raw=384, device_selection=False)
        assert bool(log.get("native_sweep")) == native and log.get("one_launch_solve") and not log.get("device_selection")
        outs.append((val, its, log["final_cost"].cpu().numpy(), log["per_restart_iterations"].cpu().numpy()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    np.testing.assert_array_equal(outs[0][3], outs[1][3])
    # a constrained sphere sweep (constraints on the sphere are user callables) stays on the Python path
    assert not run(approx=True, constrained=True, device="cuda:0", R=16, raw=64, maxiter=5)[3].get("native_sweep")


@pytest.mark.parametrize("approx,R,raw", [(False, 96, 384), (True, 512, 2048), (False, 7, 50)])
def test_native_sphere_sweep_in_one_call_with_the_selection_on_the_device(approx, R, raw, monkeypatch):
    """gabo_sphere_sweep_run: scoring -> selection kernel -> start -> solve -> arg-max with one host wait.  The restarts are distinct raw samples, and the
    Python path handed the same picks returns the same candidate, costs and iteration counts bit for bit."""
    from tools.sphere_sweep_bench import run
    from gabotorch_amd.manifold_optimization import manifold_optimize as mo
    dt, val_d, its_d, log_d = run(approx=approx, constrained=False, device="cuda:0", R=R, raw=raw, log_picked=True)
    assert log_d.get("native_sweep") and log_d.get("device_selection") and log_d.get("one_launch_solve")
    picks = log_d["picked"]
    assert picks.shape == (R,) and len(set(picks.tolist())) == R and picks.min() >= 0 and picks.max() < raw
    monkeypatch.setattr(mo, "select_rows", lambda y, n, gen, nonneg, eta=1.0, alpha=1e-4: (picks.astype(np.int64), False))
    dt, val_p, its_p, log_p = run(approx=approx, constrained=False, device="cuda:0", R=R, raw=raw, native=False)
    assert not log_p.get("native_sweep")
    assert val_d == val_p and its_d == its_p
    np.testing.assert_array_equal(log_d["final_cost"].cpu().numpy(), log_p["final_cost"].cpu().numpy())
    np.testing.assert_array_equal(log_d["per_restart_iterations"].cpu().numpy(), log_p["per_restart_iterations"].cpu().numpy())


def test_native_sphere_sweep_with_the_raw_samples_drawn_on_the_device():
    """options["device_rand"]: the raw samples of the one-call sphere sweep come from sphere_sample_kernel - the oracle's points (Philox stream, Box-Muller,
    normalised) to rounding, on the unit sphere - and the sweep over them ends where the host-sampled sweeps end."""
    import ctypes
    from oracle import selection as osel
    from tools.sphere_sweep_bench import run
    from gabotorch_amd import _lib
    lib = _lib.load()
    # the sampler alone, through the run call's workspace: count x dim points at the start of it
    dt, val, its, log = run(approx=False, constrained=False, device="cuda:0", R=64, raw=256, device_rand=True, log_picked=True)
    assert log.get("native_sweep") and log.get("device_selection") and its <= 50 and np.isfinite(val)
    from gabotorch_amd.manifold_optimization import manifold_optimize as mo
    ws = [v for k, v in mo._sweep_workspaces.items() if k[0] == "sphere"][0]
    pts = ws[:256 * 10 * 8].view(torch.float64).reshape(256, 10).cpu().numpy()          # (the raw samples sit at the start of the sweep's workspace)
    np.random.seed(5)
    seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))      # what _native_sweep_sphere drew after np.random.seed(5) in tools.sphere_sweep_bench.run
    dt0, val0, _, _ = run(approx=False, constrained=False, device="cuda:0", R=64, raw=256)
    assert abs(val - val0) < 0.05 * abs(val0)             # (the same optimum region from different raw samples)
    np.testing.assert_allclose(np.linalg.norm(pts, axis=1), 1.0, rtol=0, atol=1e-15)
    np.testing.assert_allclose(pts, osel.sphere_samples(seed, 256, 10), rtol=0, atol=1e-13)


@pytest.mark.parametrize("total,n,world", [(256, 64, 1), (2048, 512, 1), (2048, 512, 8), (50, 7, 3), (4096, 100, 1), (8192, 1, 2), (3, 1, 1)])
def test_device_selection_picks_the_oracles_rows(total, n, world):
    """gabo_spd_sweep_select_rows against oracle/selection.py (botorch's initialize_q_batch_nonneg as an exponential race on the library's Philox
    stream): the same restarts in the same order, for one rank and for every rank of a sharded table; where two keys agree to 1e-12 the order may
    differ (device exp / log against numpy's)."""
    import ctypes
    from gabotorch_amd import _lib
    from oracle import selection as osel
    lib = _lib.load()
    d, dv = 5, 15
    rng = np.random.default_rng(total + n)
    y = np.maximum(rng.standard_normal(total), 0.0) * np.exp(2.0 * rng.standard_normal(total)) + (rng.random(total) < 0.3) * 1e-3
    y[rng.integers(total)] = y.max() * 1.5
    if n > int((y > 0).sum()):
        y = np.abs(y) + 1e-6
    per = (total + world - 1) // world
    table = np.zeros((world * (per + 1), 1 + dv))
    rows = (np.arange(total) // per) * (per + 1) + 1 + np.arange(total) % per
    table[rows, 0] = y
    seed = 0x1234567 + total
    table[0, 0] = float(seed)
    want, keys = osel.select_nonneg(y, n, seed, eta=1.3, alpha=1e-4)
    assert want is not None
    tab = torch.tensor(table, device="cuda:0")
    flag = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    got_samples = None
    for rank in range(world):
        r_loc = len(range(rank, n, world))
        picked = torch.full((max((n + world - 1) // world, 1),), -1, dtype=torch.int64, device="cuda:0")
        samples = torch.full((n,), -1, dtype=torch.int64, device="cuda:0")
        rc = lib.gabo_spd_sweep_select_rows(tab.data_ptr(), d, total, per, n, 1.3, 1e-4, 0 if world > 1 else seed, 1 if world > 1 else 0, rank, world,
                                            picked.data_ptr(), samples.data_ptr(), flag.data_ptr(), None, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert int(flag) == 0
        s_ = samples.cpu().numpy()
        if got_samples is None:
            got_samples = s_
        np.testing.assert_array_equal(s_, got_samples)                   # every rank picks the same restarts
        np.testing.assert_array_equal(picked.cpu().numpy()[:r_loc], rows[s_[rank::world]])
    assert len(set(got_samples.tolist())) == n and int(np.argmax(y)) in got_samples
    differs = np.nonzero(got_samples != want)[0]
    for k in differs:              # only where the race was a photo finish
        a, b = keys[got_samples[k]], keys[want[k]]
        assert abs(a - b) <= 1e-12 * max(abs(a), abs(b)), (k, a, b)


def test_device_selection_raises_its_flag_where_the_heuristic_falls_back():
    from gabotorch_amd import _lib
    lib = _lib.load()
    for y in (-np.ones(16), np.r_[1.0, np.zeros(15)], np.r_[np.nan, np.ones(15)]):
        table = np.zeros((17, 16))
        table[1:, 0] = y
        tab = torch.tensor(table, device="cuda:0")
        flag = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        picked = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        assert lib.gabo_spd_sweep_select_rows(tab.data_ptr(), 5, 16, 16, 4, 1.0, 1e-4, 1, 0, 0, 1, picked.data_ptr(), None, flag.data_ptr(), None, None) == 0
        torch.cuda.synchronize()
        assert int(flag) == 1
    assert lib.gabo_spd_sweep_select_supported(8193, 4) == 0 and lib.gabo_spd_sweep_select_supported(16, 16) == 0


@pytest.mark.parametrize("restarts,raw", [(512, 2048), (64, 256), (7, 50)])
def test_native_sweep_with_the_selection_on_the_device(restarts, raw, monkeypatch):
    """score -> selection kernel -> solve without a host wait in between: the restarts are the oracle's picks for the scores the sweep saw, and the
    Python path handed the same picks returns the same candidate, costs and iteration counts bit for bit."""
    from tools.sweep_bench import run_sweep
    from gabotorch_amd.manifold_optimization import manifold_optimize as mo
    _, best_d, val_d, log_d = run_sweep("cuda:0", device_rand=True, builtin_constraint=True, num_restarts=restarts, raw_samples=raw, log_picked=True)
    assert log_d.get("native_sweep") and log_d.get("device_selection")
    picks = log_d["picked_samples"]
    assert picks.shape == (restarts,) and len(set(picks.tolist())) == restarts and picks.min() >= 0 and picks.max() < raw
    monkeypatch.setattr(mo, "select_rows", lambda y, n, gen, nonneg, eta=1.0, alpha=1e-4: (picks.astype(np.int64), False))
    _, best_p, val_p, log_p = run_sweep("cuda:0", device_rand=True, builtin_constraint=True, num_restarts=restarts, raw_samples=raw, native_sweep=False)
    assert not log_p.get("native_sweep")
    assert torch.equal(best_d, best_p) and val_d == val_p
    assert torch.equal(log_d["per_restart_iterations"].cpu(), log_p["per_restart_iterations"].cpu())
    np.testing.assert_array_equal(log_d["final_cost"].cpu().numpy(), log_p["final_cost"].cpu().numpy())


def test_native_sweep_falls_back_to_the_host_heuristic_when_the_selection_kernel_raises_its_flag():
    """Expected improvement that underflows to zero at every raw sample (an incumbent far below anything the surrogate predicts): botorch's heuristic
    has no positive value to weight and picks at random, with its retries and its warning (manifold_optimize.py:283-320) - the selection kernel
    raises its flag, the launches behind it do no work (skip_flag), and the sweep runs again with the heuristic on the host."""
    import functools
    import warnings
    from gabotorch_amd import manifolds, models
    from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
    from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat
    from oracle import spd as ospd
    rng = np.random.default_rng(3)
    d, n = 3, 20
    q = np.linalg.qr(rng.standard_normal((n, d, d)))[0]
    X = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.3, 3.0, (n, d)), q)
    xv = torch.tensor(ospd.symmetric_matrix_to_vector_mandel(0.5 * (X + X.transpose(0, 2, 1))), device="cuda:0")
    y = torch.tensor(rng.standard_normal(n), device="cuda:0")
    gp = models.ExactGP(xv, y, SpdAffineInvariantGaussianKernel(beta_min=0.3), outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()) - 1e4, maximize=False)
    man = manifolds.PositiveDefinite(d)
    man.min_eig, man.max_eig = 0.1, 4.0
    np.random.seed(5)
    torch.manual_seed(5)
    solver = BatchedTrustRegions(mingradnorm=1e-4, maxiter=5)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        best = joint_optimize_manifold(acq, man, solver, q=1, num_restarts=8, raw_samples=32, bounds=None,
                                       options={"device": "cuda:0", "device_rand": True},
                                       inequality_constraints=[functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=4.0)],
                                       pre_processing_manifold=to_mat, post_processing_manifold=to_vec, approx_hessian=True)
    assert solver.log.get("native_sweep") and solver.log.get("device_selection") is False
    assert any(issubclass(w.category, models.BadInitialCandidatesWarning) for w in caught)
    assert best.shape == (1, 6) and bool(torch.isfinite(best).all())
    lam = np.linalg.eigvalsh(ospd.vector_to_symmetric_matrix_mandel(best.cpu().numpy()))
    assert lam.min() > 0
