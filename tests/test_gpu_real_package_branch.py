"""The real-package branch on the device (see test_real_package_branch_cpu.py): kernels constructed on a foreign `gpytorch.kernels.Kernel` are called
through ITS __call__ and reach the HIP kernels; `joint_optimize_manifold` maximises a foreign botorch-shaped ExpectedImprovement module (an acquisition
object this package does not define: the generic lock-step path, manifold_optimize.py:36-52)."""
import pytest

from tests.test_real_package_branch_cpu import check_surface, run_branch

pytestmark = pytest.mark.gpu


def test_foreign_base_classes_reach_the_hip_kernels_and_the_maximiser():
    out = run_branch("gpu")
    check_surface(out)
    assert out["spd_forward_err"] < 1e-10 and out["sphere_forward_err"] < 1e-10
    assert abs(out["scaled_forward_ratio"] - 1.0) < 1e-12
    assert out["acq_at_optimum"] >= out["acq_at_a_training_point"] and out["acq_at_optimum"] > 0
