"""The `import gpytorch` / `import botorch` branches of gabotorch_amd._compat and gabotorch_amd.plugin_api, executed.  Neither package
exists in this image, so modules with the real packages' base-class shape (tests/_stubs/real_package_stubs.py) are put into sys.modules in a
process of its own before gabotorch_amd is imported; every kernel class of the package is then constructed on the foreign
`gpytorch.kernels.Kernel` - the registration calls of kernels_spd.py:33-70 (register_parameter / register_prior / register_constraint) run
against a base class this package does not define - and its beta / lengthscale properties round-trip through the foreign constraints."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def run_branch(*args):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_stubs", "run_real_package_branch.py"), *args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def check_surface(out):
    assert out["have_gpytorch"] and out["kernel_base_is_the_package_one"] and out["greater_than_is_the_package_one"] and out["plugin_api_reexports"]
    ks = out["kernels"]
    assert set(ks) == {"SpdAffineInvariantGaussianKernel", "SpdAffineInvariantLaplaceKernel", "SpdFrobeniusGaussianKernel", "SpdLogEuclideanGaussianKernel",
                       "NestedSpdAffineInvariantGaussianKernel", "NestedSpdLogEuclideanGaussianKernel", "SphereGaussianKernel", "SphereLaplaceKernel",
                       "NestedSphereGaussianKernel"}
    for name in ("SpdAffineInvariantGaussianKernel", "SpdAffineInvariantLaplaceKernel", "SphereGaussianKernel", "NestedSphereGaussianKernel"):
        assert "raw_beta" in ks[name]["params"] and abs(ks[name]["beta"] - 1.5) < 1e-6, ks[name]
    for name in ("SpdFrobeniusGaussianKernel", "SpdLogEuclideanGaussianKernel", "SphereLaplaceKernel"):      # (kernels_sphere.py:97-134: a lengthscale)
        assert "raw_lengthscale" in ks[name]["params"] and abs(ks[name]["lengthscale"] - 0.8) < 1e-6, ks[name]
    calls = out["registration_calls"]
    for want in ("SpdAffineInvariantGaussianKernel.register_parameter(raw_beta)", "SpdAffineInvariantGaussianKernel.register_prior(beta_prior)",
                 "SpdAffineInvariantGaussianKernel.register_constraint(raw_beta)", "SphereGaussianKernel.register_constraint(raw_beta)",
                 "ScaleKernel.register_constraint(raw_outputscale)"):
        assert want in calls, (want, calls)
    assert out["scale_kernel_params"] == ["base_kernel.raw_beta", "raw_outputscale"]


def test_kernel_classes_on_a_foreign_gpytorch_base_class():
    check_surface(run_branch())
