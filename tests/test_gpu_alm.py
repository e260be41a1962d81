"""The reference's default solver for its constrained sphere examples - AugmentedLagrangeMethod around TrustRegions
(examples/bo_sphere/constrained_benchmark_examples/gabo_sphere_equality_constraints.py:95,200-203, gabo_sphere_inequality_constraints.py:97,
238-241) - driven through `gen_candidates_manifold` as `joint_optimize_manifold` drives it (restart by restart on the host,
manifold_optimize.py:207-220), with the acquisition evaluated by the HIP kernels: every outer iterate against the record of the reference's
own classes (tests/golden/alm.npz, make_golden_alm.py).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from gabotorch_amd import manifolds, models
from gabotorch_amd.kernel_utils.kernels_sphere import SphereGaussianKernel
from gabotorch_amd.manifold_optimization.augmented_Lagrange_method import AugmentedLagrangeMethod
from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold
from gabotorch_amd.manifold_optimization.robust_trust_regions import TrustRegions
from tests.test_host_optimizers_cpu import alm_reference_walk, alm_reference_walk_batched, recording

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


@pytest.mark.parametrize("driver", ["restart_by_restart", "lock_step"])
@pytest.mark.parametrize("name", ["sph3", "sph5"])
@pytest.mark.parametrize("rname", ["eq", "ineq"])
def test_alm_through_gen_candidates_follows_the_reference_outer_iterates(golden, name, rname, driver):
    if driver == "restart_by_restart" and name == "sph5":
        pytest.skip("the host-driven form on S^4 takes a minute and adds nothing to S^2 (CPU test: tests/test_host_optimizers_cpu.py)")
    g, ga = golden("tr_traces.npz"), golden("alm.npz")
    n = int(name[3:])
    w = g[f"{name}_w"]
    kern = SphereGaussianKernel(beta_min=0.1).double()
    kern.beta = torch.tensor(float(g[f"{name}_beta"]), dtype=torch.float64)
    gp = models.ExactGP(t(g[f"{name}_Y"]), t(np.zeros(len(w))), kern, outputscale=1.0, noise=1.0, mean=0.0)
    gp._cache = (torch.eye(len(w), dtype=torch.float64, device=DEV), t(w))          # posterior mean = sum_j w_j k(x, Y_j)
    acq = models.PosteriorMean(gp, maximize=True)                                      # cost = -acq = the golden cost
    man = manifolds.Sphere(n)

    def domain_constraint(x):                       # gabo_sphere_inequality_constraints.py:112-119, on whatever device x lives
        center = torch.zeros(n, dtype=x.dtype, device=x.device)
        center[0] = 1.0
        return np.pi / 4 - torch.acos(torch.clamp((x * center).sum(), -1.0, 1.0))
    cons = dict(equality_constraints=[lambda x: x[1] - 0.0]) if rname == "eq" else dict(inequality_constraints=[domain_constraint])

    def solve_one(s, x0, rec):
        solver = AugmentedLagrangeMethod(maxiter=200, inner_solver=recording(TrustRegions(maxiter=200), rec), gammas_fact=0.05)
        c, v = gen_candidates_manifold(t(x0)[None, None], acq, man, solver, approx_hessian=False, options={"batched_alm": False}, **cons)
        x = c[0, 0].cpu().numpy()
        np.testing.assert_allclose(float(v[0]), float(acq(c[0][None]).item()), rtol=1e-12)
        return x

    def solve_all(x0s, rec):
        # the default: every restart in one batch, the fused acquisition evaluation under the penalty terms (augmented_lagrange_method.py: solve_batched)
        solver = AugmentedLagrangeMethod(maxiter=200, inner_solver=recording(TrustRegions(maxiter=200), rec), gammas_fact=0.05)
        c, v = gen_candidates_manifold(t(x0s)[:, None], acq, man, solver, approx_hessian=False, **cons)
        assert solver.log.get("batched")
        np.testing.assert_allclose(v.cpu().numpy(), acq(c).cpu().numpy(), rtol=1e-10)
        return c[:, 0].cpu().numpy(), solver.log["per_restart_iterations"].cpu().numpy()
    if driver == "lock_step":
        alm_reference_walk_batched(solve_all, ga, name, rname)
    else:
        alm_reference_walk(solve_one, ga, name, rname)
