// Acquisition value and Euclidean gradient at R candidate SPD points in ONE launch, one wave per candidate:
//
//   x*_r (Mandel)  ->  k_j = k_AI(x*_r, X_j), j = 1..n   (lane = training point j: M = L^-1 B_j L^-T, Jacobi, sum log^2)
//                  ->  GP posterior mean / variance, EI or posterior mean, d acq / d k_j      (wave-level n x n mat-vecs)
//                  ->  d acq / d x*_r = -2 L^-T [ sum_j w_j logm(M_j) ] L^-1                    (the closed form of spd_backward.hip)
//
// i.e. SpdAffineInvariant{Gaussian,Laplace}Kernel.forward (kernels_spd.py:72-100,157-187) + the exact-GP posterior and the
// analytic acquisition ([3P] botorch/gpytorch, semantics in gp_acquisition.hip) + autograd back to the candidate
// (spd_utils_torch.py:87-120), which the reference evaluates once per restart per trust-region inner iteration
// (manifold_optimize.py:175-184).  The separate-launch chain (gabo_spd_ai_pairwise -> gabo_gp_acquisition -> gabo_spd_ai_backward)
// computes the same numbers in 7 launches; this kernel exists because the lock-step trust regions are launch-count bound.
// The training side (Cholesky factors, entry-major) is prepared once per surrogate by gabo_spd_acq_prepare_train.
#include "spd_acq_kernel.hpp"

namespace gabo {

int acq_affine_invariant(const AcqLaunch& a) { return dispatch_acq<0, 12>(a); }

int acq_launch(const AcqLaunch& a) {
    switch (a.P.flags & GABO_METRIC_MASK) {
        case GABO_METRIC_AFFINE_INVARIANT: return acq_affine_invariant(a);
        case GABO_METRIC_LOG_EUCLIDEAN: return acq_log_euclidean(a);
        case GABO_METRIC_FROBENIUS: return acq_frobenius(a);
    }
    return GABO_ERR_DIM;
}

template <int D>
static int launch_prepare_train(const double* x, double* G, int64_t n, int* status, hipStream_t st) {
    launch_spd_prep<D>(nullptr, x, nullptr, G, 0, 1, 0, n, 0, 0, status, st);          // only the second (entry-major Cholesky) set
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // namespace gabo

extern "C" {

// LDS budget of the kernels built on acq_eval<D>: static tables + the trust-region tiles + 3 n doubles, within 160 KB per CU
int64_t gabo_spd_acq_max_train(int d) {
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return 0;
    const int64_t t = (int64_t)d * (d + 1) / 2;
    const int64_t fixed = (t * 64 + (d <= 8 ? 64 : (int64_t)d * d * 64) + 2 * t + 5 * d * d + 2) * 8 + 1024;
    const int64_t n = (160 * 1024 - fixed) / 24;
    return n > 2048 ? 2048 : (n < 0 ? 0 : n);
}

int gabo_spd_acq_prepare_train(const double* x_train_mandel, double* train_factors, int64_t n, int d, int* status,
                               gabo_stream_t stream) {
    if (n < 1 || n > 0x7fffffffLL || !x_train_mandel || !train_factors || !status) return GABO_ERR_ARG;
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return GABO_ERR_DIM;
#define GABO_CASE(DD) \
    case DD:          \
        return gabo::launch_prepare_train<DD>(x_train_mandel, train_factors, n, status, (hipStream_t)stream);
    switch (d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

int gabo_spd_acq_eval(const double* x_mandel, const double* train_factors, const double* alpha, const double* linv,
                      const double* linv_t, double* value, double* grad_mandel, double* scratch, int64_t r, int64_t n, int d,
                      double beta, int flags, double mean, double outputscale, double kxx, double best_f, int kind, int maximize,
                      double out_sign, const int* active, int* status, gabo_stream_t stream) {
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return GABO_ERR_DIM;
    if (r < 0 || r > 0x7fffffffLL || n < 1 || n > gabo_spd_acq_max_train(d)) return GABO_ERR_ARG;
    {
        const int out = flags & GABO_OUT_MASK, metric = flags & GABO_METRIC_MASK;
        if ((flags & ~(GABO_OUT_MASK | GABO_METRIC_MASK)) || (out != GABO_OUT_GAUSSIAN && out != GABO_OUT_LAPLACE)) return GABO_ERR_ARG;
        if (metric != GABO_METRIC_AFFINE_INVARIANT && metric != GABO_METRIC_LOG_EUCLIDEAN && metric != GABO_METRIC_FROBENIUS) return GABO_ERR_ARG;
        if (metric != GABO_METRIC_AFFINE_INVARIANT && (out != GABO_OUT_GAUSSIAN || d > 8)) return d > 8 ? GABO_ERR_DIM : GABO_ERR_ARG;
    }
    if (kind != GABO_ACQ_EXPECTED_IMPROVEMENT && kind != GABO_ACQ_POSTERIOR_MEAN) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x_mandel || !train_factors || !alpha || !value || !status) return GABO_ERR_ARG;
    if (grad_mandel && !scratch && (flags & GABO_METRIC_MASK) == GABO_METRIC_AFFINE_INVARIANT) return GABO_ERR_ARG;
    if (kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!linv || !linv_t)) return GABO_ERR_ARG;
    gabo::AcqLaunch a{x_mandel, gabo::AcqParams{train_factors, alpha, linv, linv_t, n, beta, flags, mean, outputscale, kxx, best_f, kind,
                                                maximize, out_sign},
                      value, grad_mandel, scratch, r, d, active, status, (hipStream_t)stream};
    return gabo::acq_launch(a);
}

}  // extern "C"
