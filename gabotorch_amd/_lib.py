"""ctypes binding of libgabo_hip.so (the C ABI declared in include/gabo_hip.h).

There is no fallback: if the library is missing or a call fails, this module raises.  Nothing here computes.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GABO_HIP_LIB", os.path.join(_PKG, "libgabo_hip.so"))   # override: development A/B builds only

GABO_OK = 0
GABO_ERR_DIM, GABO_ERR_ARG, GABO_ERR_NOT_SPD, GABO_ERR_LAUNCH = -1, -2, -3, -4
GABO_OUT_GAUSSIAN, GABO_OUT_DISTANCE, GABO_OUT_LAPLACE, GABO_SYMMETRIC = 0, 1, 2, 4
GABO_SPD_REG_MAX_DIM = 12
GABO_SPD_BWD_REG_MAX_DIM = 16
GABO_SPD_FWD_REG_MAX_DIM = 20
GABO_SPD_MAX_DIM = 32
GABO_TR_NESTED_MAX_DIM = 24
(GABO_SPD_EXP, GABO_SPD_LOG, GABO_SPD_INNER, GABO_SPD_NORM, GABO_SPD_DIST, GABO_SPD_EGRAD2RGRAD, GABO_SPD_EHESS2RHESS,
 GABO_SPD_LOGM, GABO_SPD_EXPM, GABO_SPD_SQRTM, GABO_SPD_EIGMAX, GABO_SPD_EIGMIN) = range(12)
GABO_ACQ_EXPECTED_IMPROVEMENT, GABO_ACQ_POSTERIOR_MEAN = 0, 1
GABO_GP_MLL_MAX_N = 160
GABO_GP_FACTOR_MAX_N = 96
GABO_GP_MLL_LARGE_MAX_N = 2048
GABO_METRIC_AFFINE_INVARIANT, GABO_METRIC_LOG_EUCLIDEAN, GABO_METRIC_FROBENIUS = 0, 8, 16
GABO_CONSTRAINT_MAX_EIGENVALUE, GABO_CONSTRAINT_MIN_EIGENVALUE = 0, 1
GABO_CONSTRAINT_MAX_EIGENVALUE_NESTED, GABO_CONSTRAINT_MIN_EIGENVALUE_NESTED = 2, 3
GABO_RECON_AFFINE_INVARIANT, GABO_RECON_LOG_EUCLIDEAN = 0, 1
GABO_RECON_MAX_LOOKAHEAD = 4
GABO_RECON_STOP = ("max iterations", "max time", "min step size", "min grad norm")     # GABO_RECON_STOP_* of the header, by code
GABO_SPH_PROJ, GABO_SPH_RETR, GABO_SPH_EXP, GABO_SPH_LOG, GABO_SPH_DIST, GABO_SPH_EHESS2RHESS = range(6)

_ERR = {GABO_ERR_DIM: "unsupported dimension", GABO_ERR_ARG: "bad argument", GABO_ERR_NOT_SPD: "input is not SPD",
        GABO_ERR_LAUNCH: "kernel launch failed"}

_c = ctypes
_P, _I64, _I, _D, _SZ = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_double, _c.c_size_t

# name -> (restype, argtypes); must list every symbol include/gabo_hip.h declares (tests/test_abi.py checks it)
class AcqParams(_c.Structure):
    """gabo_spd_acq_params of include/gabo_hip.h"""
    _fields_ = [("train_factors", _c.c_void_p), ("alpha", _c.c_void_p), ("linv", _c.c_void_p), ("linv_t", _c.c_void_p),
                ("n", _c.c_int64), ("beta", _c.c_double), ("flags", _c.c_int), ("mean", _c.c_double), ("outputscale", _c.c_double),
                ("kxx", _c.c_double), ("best_f", _c.c_double), ("kind", _c.c_int), ("maximize", _c.c_int), ("out_sign", _c.c_double)]


class SweepConfig(_c.Structure):
    """gabo_spd_sweep_config of include/gabo_hip.h"""
    _fields_ = [("acq", AcqParams), ("d", _c.c_int), ("min_eig", _c.c_double), ("max_eig", _c.c_double), ("n_constraints", _c.c_int),
                ("constraint_kind", _c.c_int * 8), ("constraint_bound", _c.c_double * 8), ("strict", _c.c_int),
                ("delta_bar", _c.c_double), ("delta0", _c.c_double), ("delta_cons", _c.c_double), ("theta", _c.c_double), ("kappa", _c.c_double),
                ("mininner", _c.c_int), ("maxinner", _c.c_int), ("rho_prime", _c.c_double), ("rho_regularization", _c.c_double),
                ("mingradnorm", _c.c_double), ("maxiter", _c.c_int64)]


class SphereAcqParams(_c.Structure):
    """gabo_sphere_acq_params of include/gabo_hip.h"""
    _fields_ = [("train", _c.c_void_p), ("train_t", _c.c_void_p), ("alpha", _c.c_void_p), ("linv", _c.c_void_p), ("linv_t", _c.c_void_p),
                ("n", _c.c_int64), ("dim", _c.c_int), ("beta", _c.c_double), ("flags", _c.c_int), ("mean", _c.c_double),
                ("outputscale", _c.c_double), ("kxx", _c.c_double), ("best_f", _c.c_double), ("kind", _c.c_int), ("maximize", _c.c_int),
                ("out_sign", _c.c_double)]


class SphereSweepConfig(_c.Structure):
    """gabo_sphere_sweep_config of include/gabo_hip.h"""
    _fields_ = [("acq", SphereAcqParams), ("delta_bar", _c.c_double), ("delta0", _c.c_double), ("theta", _c.c_double), ("kappa", _c.c_double),
                ("mininner", _c.c_int), ("maxinner", _c.c_int), ("exact_hessian", _c.c_int), ("rho_prime", _c.c_double),
                ("rho_regularization", _c.c_double), ("mingradnorm", _c.c_double), ("maxiter", _c.c_int64)]


class ReconSolveOptions(_c.Structure):
    """gabo_recon_solve_options of include/gabo_hip.h"""
    _fields_ = [(k, _c.c_double) for k in ("bound", "rho_init", "thetarho", "tau", "starting_tolgradnorm", "ending_tolgradnorm", "gammas_fact",
                                          "minstepsize", "maxtime")] + [("maxiter", _c.c_int64)] + \
               [(k, _c.c_double) for k in ("cg_minstepsize", "cg_maxtime", "cg_orth_value")] + [("cg_maxiter", _c.c_int64), ("lookahead", _c.c_int64), ("host_threads", _c.c_int64)]


class ReconSolveLog(_c.Structure):
    """gabo_recon_solve_log of include/gabo_hip.h"""
    _fields_ = [(k, _c.c_int64) for k in ("outer_iterations", "inner_iterations", "evaluations", "launches")] + [("stop_reason", _c.c_int)] + \
               [(k, _c.c_double) for k in ("violation", "rho", "gamma", "final_cost", "seconds", "seconds_evaluator")] + [("host_threads", _c.c_int64)]


# gabo_recon_eval_fn: int (void* ctx, int64 P, const double* v, c, k, double* cost, grad_v, grad_c, grad_k)
ReconEvalFn = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_int64, *([_c.POINTER(_c.c_double)] * 7))

SIGNATURES = {
    "gabo_version": (_I, []),
    "gabo_spd_ai_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I]),
    "gabo_spd_ai_pairwise": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _D, _I, _P, _SZ, _P, _P]),
    "gabo_spd_ai_backward": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _I64, _I64, _I64, _D, _I, _P, _SZ, _P, _P]),
    "gabo_spd_ai_backward2_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I]),
    "gabo_spd_ai_backward2": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _I64, _I64, _I64, _D, _I, _P, _SZ, _P, _P]),
    "gabo_nested_spd_gram_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I]),
    "gabo_nested_spd_gram": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I, _D, _I, _P, _SZ, _P, _P]),
    "gabo_sphere_pairwise": (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _D, _I, _I, _P]),
    "gabo_sphere_ktable_doubles": (_I64, []),
    "gabo_sphere_pairwise_uses_ktable": (_I, [_I64, _I64, _I64, _I, _D, _I, _I]),
    "gabo_sphere_ktable_build": (_I, [_D, _P, _P]),
    "gabo_sphere_pairwise_cached": (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _D, _I, _I, _P, _P]),
    "gabo_sphere_from_inner": (_I, [_P, _P, _I64, _D, _I, _I, _P]),
    "gabo_spd_manifold_op": (_I, [_I, _P, _P, _P, _P, _P, _P, _I64, _I, _P, _P]),
    "gabo_spd_project": (_I, [_P, _P, _P, _I64, _I, _I, _P]),
    "gabo_spd_logm_mandel": (_I, [_P, _P, _I64, _I, _P]),
    "gabo_frobenius_pairwise": (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _D, _I, _P]),
    "gabo_gp_acquisition": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _D, _D, _D, _D, _I, _I, _D, _P]),
    "gabo_gp_mll": (_I, [_P, _P, _I64, _D, _D, _D, _D, _P, _P]),
    "gabo_gp_mll_gram": (_I, [_P, _P, _I64, _D, _D, _D, _P, _P, _P]),
    "gabo_gp_factor": (_I, [_P, _P, _I64, _D, _D, _D, _P, _P, _P, _P, _P, _P]),
    "gabo_gp_mll_large_workspace_bytes": (_SZ, [_I64]),
    "gabo_gp_mll_large": (_I, [_P, _P, _I64, _D, _D, _D, _D, _I, _P, _P, _P, _SZ, _P]),
    "gabo_spd_acq_max_train": (_I64, [_I]),
    "gabo_spd_acq_prepare_train": (_I, [_P, _P, _I64, _I, _P, _P]),
    "gabo_spd_acq_eval": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _D, _I, _D, _D, _D, _D, _I, _I, _D, _P, _P, _P]),
    "gabo_spd_tcg_workspace_bytes": (_SZ, [_I64, _I, _I]),
    "gabo_spd_tcg_running_offset": (_SZ, [_I64, _I, _I]),
    "gabo_spd_tcg_begin": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _I64, _I, _I, _P, _P]),
    "gabo_spd_tcg_begin_rand": (_I, [_P, _P, _P, _I64, _I, _I, _P]),
    "gabo_spd_tcg_fd_point": (_I, [_P, _P, _I64, _I, _I, _P]),
    "gabo_spd_tcg_step": (_I, [_P, _P, _P, _I64, _I, _I, _I, _D, _D, _D, _I, _P]),
    "gabo_spd_tcg_end": (_I, [_P, _P, _P, _P, _I64, _I, _I, _P]),
    "gabo_spd_sample": (_I, [_P, _I64, _I, _D, _D, _c.c_uint64, _I, _P]),
    "gabo_spd_sample_range": (_I, [_P, _I64, _I64, _I, _D, _D, _c.c_uint64, _I, _P]),
    "gabo_nested_sphere_epilogue": (_I, [_P, _P, _I64, _I, _D, _I, _P]),
    "gabo_nested_sphere_epilogue_backward": (_I, [_P, _P, _P, _I64, _I, _D, _P]),
    "gabo_spd_tr_workspace_bytes": (_SZ, [_I64, _I, _I, _I64]),
    "gabo_spd_tr_propose": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _I64, _I, _I, _I, _D, _D, _D, _I, _I, _P, _P, _P]),
    "gabo_spd_tr_update": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I64, _D, _D, _D, _D, _I64, _P, _P]),
    "gabo_sphere_sweep_workspace_bytes": (_SZ, [_I, _I64, _I64]),
    "gabo_sphere_sweep_score": (_I, [_P, _I64, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "gabo_sphere_sweep_solve": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gabo_sphere_sweep_run": (_I, [_P, _I64, _I64, _P, _c.c_uint64, _D, _D, _c.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gabo_spd_gp_prepare_workspace_bytes": (_SZ, [_I64, _I]),
    "gabo_spd_gp_prepare": (_I, [_P, _P, _I64, _I, _D, _I, _D, _D, _D, _P, _P, _P, _P, _P, _P, _SZ, _P, _P, _P]),
    "gabo_spd_sweep_rows_workspace_bytes": (_SZ, [_I64, _I, _I64, _I64, _I]),
    "gabo_spd_sweep_rows_tables": (_I, [_P, _I64, _I, _I64, _I64, _I, _P, _P]),
    "gabo_spd_sweep_score_rows": (_I, [_P, _I64, _I64, _I64, _I64, _I64, _c.c_uint64, _P, _P, _P, _SZ, _P, _P, _I, _P]),
    "gabo_spd_sweep_select_supported": (_I, [_I64, _I64]),
    "gabo_spd_sweep_select_rows": (_I, [_P, _I, _I64, _I64, _I64, _D, _D, _c.c_uint64, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gabo_spd_sweep_solve_rows": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _SZ, _P, _P, _I, _P]),
    "gabo_spd_tr_solve": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _SZ, _I64, _I, _D, _D, _D, _I, _I, _D, _D, _D, _D, _I64,
                               _P, _P, _P, _I, _P, _P]),
    "gabo_spd_matfun_backward": (_I, [_I, _P, _P, _P, _I64, _I, _P]),
    "gabo_spd_matfun_backward_eig": (_I, [_I, _P, _P, _P, _I64, _I, _P]),
    "gabo_sphere_acq_eval": (_I, [_P, _P, _P, _P, _I64, _P]),
    "gabo_sphere_tr_workspace_bytes": (_SZ, [_I64, _I, _I]),
    "gabo_sphere_tr_stop_offset": (_SZ, [_I64, _I, _I]),
    "gabo_tr_solve_record": (_I, [_P, _I64]),
    "gabo_spd_tr_solve_supported": (_I, [_P, _I64, _I, _I, _I]),
    "gabo_spd_tr_two_waves": (_I, [_I]),
    "gabo_spd_tr_two_waves_counters": (_I, [_P, _P, _I]),
    "gabo_spd_tr_propose_supported": (_I, [_I, _I]),
    "gabo_sphere_tr_propose": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _I64, _I, _I, _D, _D, _D, _I, _I, _I, _P, _P]),
    "gabo_sphere_tr_update": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _D, _D, _D, _D, _I64, _P, _P]),
    "gabo_sphere_tr_solve": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I64, _D, _D, _I, _I, _I, _D, _D, _D, _D, _I64, _P]),
    "gabo_spd_logm_mandel_backward": (_I, [_P, _P, _P, _I64, _I, _P]),
    "gabo_frobenius_backward": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I64, _I64, _I64, _I64, _I64, _D, _I, _D, _P]),
    "gabo_sphere_manifold_op": (_I, [_I, _P, _P, _P, _P, _P, _I64, _I, _P]),
    "gabo_mandel_to_matrix": (_I, [_P, _P, _I64, _I, _P]),
    "gabo_matrix_to_mandel": (_I, [_P, _P, _I64, _I, _P]),
    "gabo_nested_spd_reconstruction_workspace_bytes": (_SZ, [_I64, _I64, _I, _I]),
    "gabo_nested_spd_reconstruction_prepare": (_I, [_P, _P, _I64, _I, _I, _P, _P]),
    "gabo_nested_spd_lift_prepare": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "gabo_nested_spd_extreme_eigenvalues": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _I, _P]),
    "gabo_nested_spd_reconstruction": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _I, _P, _SZ, _P]),
    "gabo_nested_sphere_frames": (_I, [_P, _P, _I, _I, _P]),
    "gabo_nested_sphere_project": (_I, [_P, _P, _P, _P, _P, _I64, _I, _I, _P]),
    "gabo_nested_sphere_lift": (_I, [_P, _P, _P, _P, _P, _I64, _I, _I, _P]),
    "gabo_nested_sphere_reconstruction_workspace_bytes": (_SZ, [_I64, _I64, _I, _I]),
    "gabo_nested_sphere_reconstruction": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _P, _SZ, _P]),
    "gabo_nested_sphere_fit_workspace_bytes": (_SZ, [_I64, _I, _I]),
    "gabo_nested_sphere_fit_evaluate": (_I, [_P, _P, _P, _P, _I64, _I, _I, _D, _D, _D, _D, _I, _P, _P, _SZ, _P, _SZ, _P]),
    "gabo_nested_spd_fit_workspace_bytes": (_SZ, [_I64, _I, _I]),
    "gabo_nested_spd_fit_evaluate": (_I, [_P, _P, _P, _P, _I64, _I, _I, _D, _D, _D, _D, _I, _P, _P, _SZ, _P, _SZ, _P]),
    "gabo_nested_spd_reconstruction_solve_workspace_bytes": (None, [_I64, _I, _I, _P, _P]),
    "gabo_nested_spd_reconstruction_solve": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _P, _SZ, _P, _SZ, _P, _P, _P]),
    "gabo_nested_spd_reconstruction_solve_with": (_I, [ReconEvalFn, _P, _P, _P, _P, _P, _P, _I, _I, _P, _SZ, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built.  Run `python -m gabotorch_amd._build` "
                "(needs hipcc).  gabotorch_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64.so.7; the library must bind to THAT runtime (the streams and pointers it is handed
        # belong to it).  Same SONAME as the system one, so importing torch first is enough - loading this library before torch
        # would pull in /opt/rocm's copy and leave two HIP runtimes in the process (every launch then fails).
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header and library out of sync
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(code, what):
    if code != GABO_OK:
        raise RuntimeError(f"{what}: libgabo_hip error {code} ({_ERR.get(code, 'unknown')})")
