"""`beta_float()` of the SPD kernel classes - the launch argument of the fused paths - carries the bits of `float(kernel.beta.double())`, the
gpytorch-style property of the reference (kernels_spd.py:33-70: softplus of the raw parameter plus the constraint's lower bound), for fp32
parameters (the default) and for a module cast to double (the trust-region fixtures do that)."""
import warnings

import pytest
import torch

from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel, SpdAffineInvariantLaplaceKernel


@pytest.mark.parametrize("double", [False, True])
@pytest.mark.parametrize("cls", [SpdAffineInvariantGaussianKernel, SpdAffineInvariantLaplaceKernel])
def test_beta_float_has_the_bits_of_the_property(cls, double):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for beta_min in (0.25, 0.1, 1e-3, 3.7):
            for raw in (0.0, -3.2, 0.333, 1.7, 25.0, -30.0):
                k = cls(beta_min=beta_min)
                if double:
                    k = k.double()
                with torch.no_grad():
                    k.raw_beta.fill_(raw)
                assert k.beta_float() == float(k.beta.double())
                assert k.beta_float() == float(k.beta.double())            # (remembered per raw value)
                with torch.no_grad():
                    k.raw_beta.add_(0.125)
                assert k.beta_float() == float(k.beta.double())            # ... and recomputed when it changes
            k = cls(beta_min=beta_min)
            if double:
                k = k.double()
            k.beta = torch.tensor(beta_min + 0.7312345678901, dtype=torch.float64 if double else torch.float32)
            assert k.beta_float() == float(k.beta.double())
