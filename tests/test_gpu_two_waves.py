"""The two-wave form of the single-launch trust-region solve (csrc/spd_tr_duo_body.hpp) against the one-wave form: the same results bit for bit, for every
dimension it is built for, with and without eigenvalue bounds, strict and not, with the bit-identical shortcuts on and off; and that the form was the one that
ran (its speculation counters move).  The toggle is gabo_spd_tr_two_waves (include/gabo_hip.h)."""
import ctypes
import functools

import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, manifolds, models, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel
from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold
from gabotorch_amd.Riemannian_utils import spd_constraints_utils_torch as scut
from gabotorch_amd.Riemannian_utils.spd_utils_torch import (symmetric_matrix_to_vector_mandel_torch,
                                                            vector_to_symmetric_matrix_mandel_torch)
from oracle import spd as ospd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=DEV)


def _problem(d, n_train, seed, R):
    rng = np.random.default_rng(seed)
    q = np.linalg.qr(rng.standard_normal((n_train, d, d)))[0]
    Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.2, 3.0, (n_train, d)), q)
    X = ospd.symmetric_matrix_to_vector_mandel(0.5 * (Xm + Xm.transpose(0, 2, 1)))
    y = np.log(np.linalg.eigvalsh(Xm)).sum(1) ** 2 + 0.1 * rng.standard_normal(n_train)
    gp = models.ExactGP(t(X), t(y), SpdAffineInvariantGaussianKernel(beta_min=0.5), outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=float(y.min()), maximize=False)
    q = np.linalg.qr(rng.standard_normal((R, d, d)))[0]
    P = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.4, 2.4, (R, d)), q)
    x0 = ops.matrix_to_mandel(t(0.5 * (P + P.transpose(0, 2, 1))))[:, None]
    return acq, x0


def _counters(lib, reset):
    h, m = ctypes.c_longlong(0), ctypes.c_longlong(0)
    _lib.check(lib.gabo_spd_tr_two_waves_counters(ctypes.byref(h), ctypes.byref(m), 1 if reset else 0), "gabo_spd_tr_two_waves_counters")
    return h.value, m.value


def _solve(acq, x0, d, cons, strict, maxiter):
    solver = BatchedTrustRegions(mingradnorm=1e-5, maxiter=maxiter, strict_constraints=strict)
    c, v = gen_candidates_manifold(x0, acq, manifolds.PositiveDefinite(d), solver, vector_to_symmetric_matrix_mandel_torch,
                                   symmetric_matrix_to_vector_mandel_torch, inequality_constraints=cons, approx_hessian=True, options={})
    assert "one_launch_solve" in solver.log
    return c.cpu().numpy(), v.cpu().numpy(), solver.log["per_restart_iterations"].cpu().numpy()


@pytest.mark.parametrize("case", ["unconstrained", "max_eig", "box", "box_strict"])
@pytest.mark.parametrize("d", [2, 3, 4, 5, 6])
def test_two_waves_match_one_wave_bit_for_bit(d, case):
    lib = _lib.load()
    acq, x0 = _problem(d, n_train=20 if d < 6 else 14, seed=300 + d, R=48)
    cons = None
    if case != "unconstrained":
        cons = [functools.partial(scut.max_eigenvalue_constraint_torch, maximum_eigenvalue=2.5)]
        if case.startswith("box"):
            cons.append(functools.partial(scut.min_eigenvalue_constraint_torch, minimum_eigenvalue=0.35))
    strict = case.endswith("strict")
    before = lib.gabo_spd_tr_two_waves(-1)
    ops.set_error_checking(False)
    try:
        lib.gabo_spd_tr_two_waves(1)
        _counters(lib, reset=True)
        two = _solve(acq, x0, d, cons, strict, maxiter=25)
        hits, misses = _counters(lib, reset=True)
        lib.gabo_spd_tr_two_waves(0)
        one = _solve(acq, x0, d, cons, strict, maxiter=25)
        assert _counters(lib, reset=True) == (0, 0)
    finally:
        lib.gabo_spd_tr_two_waves(before)
        ops.set_error_checking(True)
    assert hits + misses >= int(two[2].sum()) > 0 or hits + misses > 0          # the two-wave kernel ran (iterations applied as scalar updates are not counted)
    assert hits > 0
    for a, b in zip(two, one):
        np.testing.assert_array_equal(a, b)


def test_two_waves_decline_what_they_are_not_built_for():
    """More restarts than there are SIMDs for two waves each, and d = 7: the one-wave kernel runs (counters stay), same entry point."""
    lib = _lib.load()
    before = lib.gabo_spd_tr_two_waves(1)
    ops.set_error_checking(False)
    try:
        for d, R in ((3, 600), (7, 16)):
            acq, x0 = _problem(d, n_train=12, seed=7 + d, R=R)
            _counters(lib, reset=True)
            _solve(acq, x0, d, None, False, maxiter=4)
            assert _counters(lib, reset=True) == (0, 0)
    finally:
        lib.gabo_spd_tr_two_waves(before)
        ops.set_error_checking(True)
