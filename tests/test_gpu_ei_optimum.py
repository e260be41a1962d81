"""north_star's literal statement - "acquisition optima within 1e-5" - for the COMPOSED result: the candidate `joint_optimize_manifold`
returns (arg max over the restarts, manifold_optimize.py:118-120) against the reference's best-of-restarts from the same initial conditions.

tests/golden/ei_optimum.npz (make_golden_ei_optimum.py): the reference's own `ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100)` +
`get_hessianfd`, run restart by restart on a config-4-sized EXPECTED-IMPROVEMENT problem (exact GP on 50 Ackley observations on S^5_++,
lambda_max <= 5 constraint, 32 starts with the raw samples' distribution, eight of them on the constraint's edge), with its cost stated on
the reference's `affine_invariant_distance_torch`.  Here: the same surrogate through the HIP kernels, every execution plan of the maximiser."""
import functools

import numpy as np
import pytest
import torch

from gabotorch_amd import manifolds, models, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.manifold_optimization import manifold_optimize as mo
from gabotorch_amd.manifold_optimization.constrained_trust_regions import ConstrainedTrustRegions
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch as to_vec,
                                                            vector_to_symmetric_matrix_mandel_torch as to_mat)
from oracle import gp as ogp
from oracle import spd as ospd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(g):
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)      # noqa: E731
    kern = SpdAffineInvariantGaussianKernel(beta_min=0.25)                # raw_beta = 0: beta = 0.25 + ln 2 (fp32 softplus, then double)
    gp = models.ExactGP(t(g["Xv"]), t(g["y"]), kern, outputscale=float(g["outputscale"]), noise=float(g["noise"]), mean=float(g["mean"]))
    acq = models.ExpectedImprovement(gp, best_f=float(g["best_f"]), maximize=False)
    man = manifolds.PositiveDefinite(5)
    man.min_eig, man.max_eig = 1e-3, 5.0
    return gp, acq, man


PLANS = {"single_launch_solve": ("partial", {}),
         "propose_update_launches": ("partial", {"device_solve": False}),
         "propose_update_hipgraphs": ("opaque", {"hip_graphs": True}),
         "device_tcg_only": ("opaque", {"device_iteration": False}),
         "generic_lockstep_fused_evaluations": ("opaque", {"device_tcg": False, "device_outer": False}),
         "generic_lockstep_autograd": ("opaque", {"fused_acquisition": False})}


def test_surrogate_matches_the_reference_cost_at_the_starts_and_optima(golden):
    """-EI of the fixture (torch on the reference's distance function) = the acquisition through the HIP kernels, at the 32 starts and at the
    reference's 32 end points; and the numpy oracle agrees (botorch's EI formula is restated in all three places: [3P], unpinned)."""
    g = golden("ei_optimum.npz")
    gp, acq, man = _problem(g)
    beta = float(gp.base_kernel.beta)
    assert abs(beta - float(g["beta"])) < 1e-7                           # the kernel's fp32 softplus(0) + beta_min against 0.25 + ln 2 in double
    gp.base_kernel.beta = float(g["beta"])
    for pts, want in ((g["x0"], g["f0"]), (g["x"], g["f"])):
        v = ospd.symmetric_matrix_to_vector_mandel(pts)
        with torch.no_grad():
            got = -acq(torch.tensor(v, device=DEV)[:, None]).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-12)     # (the fixture differentiates k(x, x) too; values agree to rounding)
        ks = ospd.spd_ai_gaussian_kernel(v, g["Xv"], float(g["beta"]))
        kxx = ospd.spd_ai_gaussian_kernel(g["Xv"], g["Xv"], float(g["beta"]))
        ora = ogp.expected_improvement(*ogp.gp_posterior(kxx, ks, np.exp(-float(g["beta"]) * 1e-15) * np.ones(len(v)), g["y"], float(g["mean"]),
                                                         float(g["outputscale"]), float(g["noise"])), best_f=float(g["best_f"]), maximize=False)
        np.testing.assert_allclose(-ora, want, rtol=2e-7, atol=1e-12)


@pytest.mark.parametrize("plan", list(PLANS))
def test_returned_candidate_is_the_reference_best_of_restarts(plan, monkeypatch, golden):
    g = golden("ei_optimum.npz")
    gp, acq, man = _problem(g)
    gp.base_kernel.beta = float(g["beta"])
    R = g["x0"].shape[0]
    ic = to_vec(torch.tensor(g["x0"], device=DEV))[:, None]                   # R x 1 x 15, what gen_batch_initial_conditions_manifold returns
    monkeypatch.setattr(mo, "gen_batch_initial_conditions_manifold", lambda **kw: ic.clone())
    kind, opts = PLANS[plan]
    cons = [functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=5.0)] if kind == "partial" \
        else [lambda m: scut.max_eigenvalue_constraint_torch(m, 5.0)]
    ops.set_error_checking(False)
    solver = ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100)
    best = mo.joint_optimize_manifold(acq, man, solver, q=1, num_restarts=R, raw_samples=4 * R, bounds=None,
                                      options=dict(opts, device=DEV), inequality_constraints=cons, pre_processing_manifold=to_mat,
                                      post_processing_manifold=to_vec, approx_hessian=True)
    ops.set_error_checking(True)
    assert best.shape == (1, 15)
    with torch.no_grad():
        f_star = -float(acq(best[None]).item())
    ref_best = int(g["best"])
    f_ref = float(g["f"][ref_best])
    # THE statement: the returned optimum against the reference's best of restarts: 1e-8 relative (north_star asks for 1e-5; 2e-9 is what the
    # reference's own evaluation of EI differs from every plan here by, the autograd path included)
    assert abs(f_star - f_ref) <= 1e-8 * abs(f_ref), (plan, f_star, f_ref)
    # and the candidate itself is the reference's
    x_star = ospd.vector_to_symmetric_matrix_mandel(best.cpu().numpy())[0]
    assert np.linalg.norm(x_star - g["x"][ref_best]) <= 1e-7 * np.linalg.norm(g["x"][ref_best]), plan

    # restart by restart (the values behind the arg max): restarts the reference ends by its gradient / step criteria within 1e-8 relative;
    # the one it stops at maxiter - crawling along the bound, |f| = 6e-4 - within 1e-7 (rounds 3-4: 1e-5 and 2e-3)
    cands, vals = mo.gen_candidates_manifold(ic, acq, man, ConstrainedTrustRegions(mingradnorm=1e-4, maxiter=100), to_mat, to_vec,
                                             inequality_constraints=cons, approx_hessian=True, options=opts)
    f = -vals.cpu().numpy()
    conv = g["nit"] < 100
    assert conv.sum() >= R - 2
    np.testing.assert_allclose(f[conv], g["f"][conv], rtol=1e-8, atol=0, err_msg=plan)
    np.testing.assert_allclose(f[~conv], g["f"][~conv], rtol=1e-7, atol=0, err_msg=plan)
    assert int(np.argmin(f)) == int(np.argmin(np.where(np.isclose(g["f"], f_ref, rtol=1e-9), g["f"], np.inf))) or \
        abs(f.min() - f_ref) <= 1e-5 * abs(f_ref)
