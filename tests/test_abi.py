"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares."""
import os
import re

import pytest

from gabotorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    """The .so is a build artefact (git-ignored): build it with hipcc when a fresh checkout has none."""
    if not os.path.exists(_lib.LIB_PATH):
        from gabotorch_amd import _build
        _build.build()


def _declared():
    text = open(os.path.join(ROOT, "include", "gabo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(gabo_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert names, "no declarations parsed from include/gabo_hip.h"
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gabo_hip.h but not exported by libgabo_hip.so"
    assert names == set(_lib.SIGNATURES), (names ^ set(_lib.SIGNATURES))


def test_host_side_argument_errors_need_no_gpu():
    lib = _lib.load()
    assert lib.gabo_version() >= 100
    assert lib.gabo_spd_ai_workspace_bytes(2, 3, 4, 10) == 2 * 7 * 55 * 8
    assert lib.gabo_spd_ai_workspace_bytes(2, 3, 4, 24) == 2 * 2 * 3 * 576 * 8      # wave-per-pair fallback layout
    # argument validation happens before any HIP call
    assert lib.gabo_spd_ai_pairwise(None, None, None, None, 1, 4, 4, 40, 0, 0, 1.0, 0, None, 0, None, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_spd_ai_pairwise(None, None, None, None, 1, 4, 4, 3, 0, 0, 1.0, 0, None, 0, None, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_ai_pairwise(None, None, None, None, 1, 0, 4, 3, 0, 0, 1.0, 0, None, 0, None, None) == _lib.GABO_OK
    assert lib.gabo_sphere_pairwise(None, None, None, 1, 4, 4, 0, 0, 0, 1.0, 0, 0, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_mandel_to_matrix(None, None, 3, 65, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_mandel_to_matrix(None, None, 0, 5, None) == _lib.GABO_OK


def test_acquisition_and_trust_region_entry_points_validate_on_the_host():
    import ctypes
    lib = _lib.load()
    # fused GP acquisition: n out of range, unknown kind, empty batch
    assert lib.gabo_gp_acquisition(None, None, None, None, None, None, 4, 0, 0.0, 1.0, 1.0, 0.0, 0, 1, 1.0, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_gp_acquisition(None, None, None, None, None, None, 4, 8, 0.0, 1.0, 1.0, 0.0, 7, 1, 1.0, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_gp_acquisition(None, None, None, None, None, None, 0, 8, 0.0, 1.0, 1.0, 0.0, 0, 1, 1.0, None) == _lib.GABO_OK
    # single-launch SPD acquisition: d > 12 is served by the separate-launch chain, not by this entry
    assert lib.gabo_spd_acq_eval(None, None, None, None, None, None, None, None, 4, 8, 13, 1.0, 0, 0.0, 1.0, 1.0, 0.0, 0, 1, 1.0, None, None, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_spd_acq_eval(None, None, None, None, None, None, None, None, 4, 8, 5, 1.0, _lib.GABO_OUT_DISTANCE, 0.0, 1.0, 1.0, 0.0, 0, 1, 1.0, None, None, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_acq_prepare_train(None, None, 8, 5, None, None) == _lib.GABO_ERR_ARG
    # tCG / trust-region state machines: workspace sizes grow with every argument, bad dims and constraint counts are refused
    w0 = lib.gabo_spd_tcg_workspace_bytes(16, 5, 0)
    assert 0 < w0 < lib.gabo_spd_tcg_workspace_bytes(16, 5, 2) < lib.gabo_spd_tcg_workspace_bytes(17, 5, 2)
    assert 0 < lib.gabo_spd_tcg_running_offset(16, 5, 2) < lib.gabo_spd_tcg_workspace_bytes(16, 5, 2)
    assert lib.gabo_spd_tr_workspace_bytes(16, 5, 2, 50) > lib.gabo_spd_tcg_workspace_bytes(16, 5, 2) + 16 * 15 * 50 * 8
    assert lib.gabo_spd_tcg_begin(None, None, None, None, None, None, None, 0, 4, 33, 0, None, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_spd_tcg_begin(None, None, None, None, None, None, None, 0, 4, 5, 9, None, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_tcg_step(None, None, None, 4, 5, 2, 3, 1e-6, 1.0, 0.1, 1, None) == _lib.GABO_ERR_ARG    # n_eq > n_constraints
    assert lib.gabo_spd_tcg_end(None, None, None, None, 0, 5, 0, None) == _lib.GABO_OK
    acq = _lib.AcqParams()
    assert lib.gabo_spd_tr_propose(None, None, None, None, None, None, ctypes.byref(acq), None, 0, None, 4, 13, 0, 0, 1e-6, 1.0, 0.1, 1, 15, None, None, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_spd_tr_propose(None, None, None, None, None, None, None, None, 0, None, 4, 5, 0, 0, 1e-6, 1.0, 0.1, 1, 15, None, None, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_tr_update(None, None, None, None, None, None, None, None, None, None, 0, 5, 0, 50, 3.0, 0.1, 1e3, 1e-4, 100, None, None) == _lib.GABO_OK
    # gradients of the Frobenius / log-Euclidean kernels
    assert lib.gabo_frobenius_backward(None, None, None, None, 1, 4, 4, 33, 0, 0, 16, 4, 1, 1.0, 0, 1.0, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_spd_logm_mandel_backward(None, None, None, 4, 5, None) == _lib.GABO_ERR_ARG


def test_workspace_covers_both_layouts_between_the_register_limits():
    """12 < d <= 16: the forward kernels use the packed layout, the backward the wave-per-pair one - the workspace must fit either."""
    lib = _lib.load()
    assert _lib.GABO_SPD_REG_MAX_DIM == 12 and _lib.GABO_SPD_FWD_REG_MAX_DIM == 20 and _lib.GABO_SPD_BWD_REG_MAX_DIM == 16
    for d in (13, 16):
        t = d * (d + 1) // 2
        assert lib.gabo_spd_ai_workspace_bytes(1, 3, 50, d) == max((3 + 50) * t, 2 * 3 * d * d) * 8
        assert lib.gabo_spd_ai_workspace_bytes(1, 50, 3, d) == max((3 + 50) * t, 2 * 50 * d * d) * 8
    assert lib.gabo_spd_ai_workspace_bytes(1, 3, 50, 21) == 2 * 3 * 21 * 21 * 8


def test_round3_host_loops_and_nested_sphere_entries_validate_on_the_host():
    """The entry points added for the HD-GaBO loops: argument validation and workspace sizes need no GPU."""
    import ctypes
    lib = _lib.load()
    # nested-sphere chain
    assert lib.gabo_nested_sphere_frames(None, None, 1, 1, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_nested_sphere_frames(None, None, 5, 5, None) == _lib.GABO_ERR_DIM            # levels <= D - 1
    assert lib.gabo_nested_sphere_frames(None, None, 5, 2, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_nested_sphere_project(None, None, None, None, None, 0, 5, 2, None) == _lib.GABO_OK
    assert lib.gabo_nested_sphere_project(None, None, None, None, None, 3, 5, 2, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_nested_sphere_lift(None, None, None, None, None, 3, 5, 0, None) == _lib.GABO_ERR_DIM
    w1 = lib.gabo_nested_sphere_reconstruction_workspace_bytes(1, 10, 21, 18)
    assert 0 < w1 < lib.gabo_nested_sphere_reconstruction_workspace_bytes(3, 10, 21, 18) and w1 >= 16 + 10 * 19 * 8
    assert lib.gabo_nested_sphere_reconstruction(None, None, None, None, None, None, 0, 4, 21, 18, None, 0, None) == _lib.GABO_OK
    assert lib.gabo_nested_sphere_reconstruction(None, None, None, None, None, None, 1, 4, 21, 18, None, 0, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_nested_sphere_reconstruction(None, None, None, None, None, None, 1, 4, 300, 298, None, 0, None) == _lib.GABO_ERR_DIM   # LDS
    assert lib.gabo_nested_sphere_fit_workspace_bytes(12, 51, 48) > 12 * (51 * 48 - 48 * 47 // 2) * 8
    assert lib.gabo_nested_sphere_fit_evaluate(None, None, None, None, 12, 51, 48, 1.0, 1.0, 0.1, 0.0, 1, None, None, 0, None, 0, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_nested_sphere_fit_evaluate(None, None, None, None, 12, 80, 10, 1.0, 1.0, 0.1, 0.0, 1, None, None, 0, None, 0, None) == _lib.GABO_ERR_DIM  # latent > 64
    # nested-SPD fit evaluation and the native reconstruction loop
    assert lib.gabo_nested_spd_fit_workspace_bytes(12, 20, 2) > 3 * 12 * 12 * 8
    assert lib.gabo_nested_spd_fit_evaluate(None, None, None, None, 12, 40, 2, 1.0, 1.0, 0.1, 0.0, 1, None, None, 0, None, 0, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_nested_spd_fit_evaluate(None, None, None, None, 12, 20, 2, 1.0, 1.0, 0.1, 0.0, 1, None, None, 0, None, 0, None) == _lib.GABO_ERR_ARG
    dev, pin = ctypes.c_size_t(0), ctypes.c_size_t(0)
    lib.gabo_nested_spd_reconstruction_solve_workspace_bytes(10, 20, 2, ctypes.byref(dev), ctypes.byref(pin))
    npar = 20 * 18 + 18 * 18 + 2 * 18
    assert pin.value == _lib.GABO_RECON_MAX_LOOKAHEAD * (2 * npar + 1 + 18 + 18 * 18) and dev.value > pin.value * 8
    opts = _lib.ReconSolveOptions(bound=20, rho_init=1, thetarho=0.3, tau=0.8, starting_tolgradnorm=1e-3, ending_tolgradnorm=1e-6, gammas_fact=1.0,
                                  minstepsize=1e-10, maxtime=10, maxiter=2, cg_minstepsize=1e-10, cg_maxtime=10, cg_orth_value=1e300, cg_maxiter=3)
    log = _lib.ReconSolveLog()
    assert lib.gabo_nested_spd_reconstruction_solve(None, None, None, None, None, None, None, None, None, 4, 40, 2, 1, None, 0, None, 0,
                                                    ctypes.byref(opts), ctypes.byref(log), None) == _lib.GABO_ERR_DIM
    assert lib.gabo_nested_spd_reconstruction_solve(None, None, None, None, None, None, None, None, None, 4, 20, 2, 1, None, 0, None, 0,
                                                    ctypes.byref(opts), ctypes.byref(log), None) == _lib.GABO_ERR_ARG
    assert lib.gabo_nested_spd_reconstruction(None, None, None, None, None, None, None, None, None, None, None, None, None, 0, 4, 20, 2, 1, None, 0,
                                              None) == _lib.GABO_OK
