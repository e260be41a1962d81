"""End-to-end BO loops (BASELINE.json configs 1 and 4 in miniature): GP fit + EI + manifold maximiser on the HIP path."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
pytestmark = pytest.mark.gpu


def test_gabo_sphere_loop():
    import gabo_sphere
    x, y, best = gabo_sphere.run(dim=3, iters=8, verbose=False)
    assert x.shape == (13, 3) and np.allclose(x.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-12)
    assert all(b2 <= b1 + 1e-12 for b1, b2 in zip(best, best[1:])) and np.isfinite(best[-1])
    assert best[-1] < best[0]            # EI finds something better than 5 random points on S^2 within 8 iterations


def test_gabo_spd_loop():
    import gabo_spd
    from oracle import spd as ospd
    x, y, best = gabo_spd.run(dim=3, iters=6, verbose=False)
    assert x.shape == (11, 6)
    lam = np.linalg.eigvalsh(ospd.vector_to_symmetric_matrix_mandel(x.cpu().numpy()))
    assert lam.min() > 0 and lam.max() < 5.5
    assert all(b2 <= b1 + 1e-12 for b1, b2 in zip(best, best[1:])) and np.isfinite(best[-1])


def test_hd_gabo_spd_loop():
    """Config-5-shaped flow in miniature: S^5_++ observations, nested projection to S^2_++, latent EI maximisation, lift back."""
    import hd_gabo_spd
    from oracle import spd as ospd
    x, y, best = hd_gabo_spd.run(dim=5, latent=2, iters=4, verbose=False)
    assert x.shape == (9, 15)
    lam = np.linalg.eigvalsh(ospd.vector_to_symmetric_matrix_mandel(x.cpu().numpy()))
    assert lam.min() > 0
    assert all(b2 <= b1 + 1e-12 for b1, b2 in zip(best, best[1:])) and np.isfinite(best[-1])


def test_hd_gabo_sphere_loop():
    """HD-GaBO on S^4 through nested spheres down to S^2: axes learnt on their spheres, reconstruction distances optimised,
    latent EI maximisation, lift back."""
    import hd_gabo_sphere
    x, y, best = hd_gabo_sphere.run(dim=5, latent=3, iters=3, verbose=False, fit_iters=15)
    assert x.shape == (8, 5)
    np.testing.assert_allclose(np.linalg.norm(x.cpu().numpy(), axis=1), 1.0, atol=1e-12)
    assert all(b2 <= b1 + 1e-12 for b1, b2 in zip(best, best[1:])) and np.isfinite(best[-1])


@pytest.mark.parametrize("kind,solver", [("equality", None), ("equality", "CTR"), ("inequality", None), ("bounds", None)])
def test_constrained_sphere_loops(kind, solver):
    """The reference's constrained sphere examples (examples/bo_sphere/constrained_benchmark_examples/*.py) in miniature: their constraint
    callables, their constrained `manifold.rand`, their solvers (AugmentedLagrangeMethod around TrustRegions by default, ConstrainedTrustRegions
    for the bounds).  Every candidate the maximiser returns satisfies the constraints (to the method's tolerance)."""
    import gabo_sphere_constraints as ex
    x, y, best, feasible = ex.run(kind, solver, iters=3, verbose=False, alm_maxiter=60)
    xs = x.cpu().numpy()
    assert xs.shape == (8, 3) and np.allclose(np.linalg.norm(xs, axis=1), 1.0, atol=1e-12)
    assert all(feasible(p) for p in xs), [p for p in xs if not feasible(p)]
    assert all(b2 <= b1 + 1e-12 for b1, b2 in zip(best, best[1:])) and np.isfinite(best[-1])
