"""The config-4 sweep as two native host calls (gabo_spd_sweep_score / gabo_spd_sweep_solve, csrc/spd_sweep.hip) against the Python path of
joint_optimize_manifold (manifold_optimize.py:36-120 of the reference: initial conditions, the restarts' solves, argmax): the native driver
enqueues the same launches in the same order and leaves the selection heuristic and both random generators where they are, so the returned
candidate is the same BIT FOR BIT."""
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
    _, best_n, val_n, log_n = run_sweep("cuda:0", device_rand=device_rand, builtin_constraint=True, native_sweep=True, **kw)
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
                                       options={"device": "cuda:0", "batched_rand": True, "native_sweep": native}, inequality_constraints=cons,
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
        dt, val, its, log = run(approx=approx, constrained=False, device="cuda:0", native=native, R=96, raw=384)
        assert bool(log.get("native_sweep")) == native and log.get("one_launch_solve")
        outs.append((val, its, log["final_cost"].cpu().numpy(), log["per_restart_iterations"].cpu().numpy()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    np.testing.assert_array_equal(outs[0][3], outs[1][3])
    # a constrained sphere sweep (constraints on the sphere are user callables) stays on the Python path
    assert not run(approx=True, constrained=True, device="cuda:0", R=16, raw=64, maxiter=5)[3].get("native_sweep")
