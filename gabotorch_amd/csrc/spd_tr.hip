// One Riemannian trust-region iteration of the acquisition maximiser on S^d_++ for R restarts, one wave per restart, in two
// launches:
//   gabo_spd_tr_propose:  tCG (begin, up to maxinner x [FD point -> acquisition gradient -> step]), proposal
//                         x+ = retr(x, eta) = L expm(eta~) L^T, acquisition value and gradient at x+, model decrease.
//   gabo_spd_tr_update:   rho test, radius update, accept / reject, new Riemannian gradient and its norm, stopping flags.
// Reference: TrustRegions.solve / ConstrainedTrustRegions.solve / StrictConstrainedTrustRegions.solve
// (robust_trust_regions.py:111-415, constrained_trust_regions.py:120-528,737-1160) driving one restart at a time with the
// acquisition behind pymanopt_addons/problem.py; here restarts never interact, so each wave runs its whole inner loop alone
// and the host only supplies what is user code: the constraint callables, evaluated between the two launches when strict.
// The building blocks are the device functions of spd_tcg_body.hpp and spd_acq_body.hpp.
#include <cstdlib>

#include "spd_tr_body.hpp"

namespace gabo {

__global__ __launch_bounds__(64) void spd_tr_update_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                           double* __restrict__ ng, double* __restrict__ delta_tr,
                                                           uint8_t* __restrict__ active, int64_t* __restrict__ iters,
                                                           const uint8_t* __restrict__ invalid, const double* __restrict__ x_prop,
                                                           void* wsbase, int64_t R, int d, int C, int64_t n, double delta_bar,
                                                           double rho_prime, double rho_regularization, double mingradnorm,
                                                           int64_t maxiter, int* __restrict__ any_active) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    const int64_t i = blockIdx.x;
    if (active[i] == 0) return;
    TrWs t = tr_layout(wsbase, R, d, C, n);
    const bool inval = invalid != nullptr && invalid[i] != 0;
    const bool still = tr_update_body(x + i * dd, fx + i, g + i * dd, ng + i, delta_tr + i, iters + i, inval, x_prop + i * dd, t, i, d, C,
                                      delta_bar, rho_prime, rho_regularization, mingradnorm, maxiter, lds);
    if (threadIdx.x == 0) {
        if (!still) active[i] = 0;
        else atomicOr(any_active, 1);
    }
}

int propose_affine_invariant(const ProposeArgs& a) { return a.d <= 8 ? dispatch_propose<0, 8>(a) : propose_affine_invariant_wide(a); }

}  // namespace gabo

namespace gabo {
// gabo_tr_solve_record: one pending buffer per host THREAD, consumed by that thread's next gabo_spd_tr_solve / gabo_sphere_tr_solve (a parity /
// debugging facility).  The sweep drivers (spd_sweep.hip) launch their solves through tr_solve_dispatch and never take it.
static thread_local double* g_tr_record = nullptr;
static thread_local int64_t g_tr_record_cap = 0;
void tr_record_take(double** buffer, int64_t* capacity) {
    *buffer = g_tr_record;
    *capacity = g_tr_record_cap;
    g_tr_record = nullptr;
    g_tr_record_cap = 0;
}

int tr_solve_dispatch(const SolveArgs& a0) {
    SolveArgs a = a0;
    // test hook: GABO_TR_NO_SHORTCUTS in the environment runs every iteration in full (no value-first evaluation after a rejection, no reuse of
    // an identical step's proposal): the two forms must agree bit for bit (tests/test_gpu_native_sweep.py).  Read once per process.
    static const int shortcuts = getenv("GABO_TR_NO_SHORTCUTS") ? 0 : 1;
    a.shortcuts = shortcuts;
    switch (a.P->flags & GABO_METRIC_MASK) {
        case GABO_METRIC_AFFINE_INVARIANT: return solve_affine_invariant(a);
        case GABO_METRIC_LOG_EUCLIDEAN: return solve_log_euclidean(a);
        case GABO_METRIC_FROBENIUS: return solve_frobenius(a);
    }
    return GABO_ERR_ARG;
}

bool tr_solve_uses_global_workspace(const AcqParams& P, int64_t r, int d, int C, size_t nested_bytes) {
    int stage_gp = 0, ws_lds = 0;
    tr_solve_dynamic_lds(P.n, r, d, C, &stage_gp, &ws_lds, nested_bytes, nullptr, tr_factor_count(P));
    return !ws_lds;
}
}  // namespace gabo

extern "C" {

size_t gabo_spd_tr_workspace_bytes(int64_t r, int d, int n_constraints, int64_t n_train) {
    if (r < 0 || d < 1 || n_constraints < 0 || n_train < 0) return 0;
    return gabo::tr_layout(nullptr, r, d, n_constraints, n_train).bytes;
}

int gabo_spd_tr_propose_supported(int flags, int d) {
    const int metric = flags & GABO_METRIC_MASK;
    if (metric == GABO_METRIC_AFFINE_INVARIANT) return d >= 2 && d <= GABO_SPD_REG_MAX_DIM;
    if (metric == GABO_METRIC_LOG_EUCLIDEAN) return d >= 2 && d <= (GABO_LE_MAX_GENERIC_DIM >= 8 ? 8 : 7);       /* (d = 8: spd_tr_le_hi.hip) */
    if (metric == GABO_METRIC_FROBENIUS) return d >= 2 && d <= 8;
    return 0;
}

int gabo_spd_tr_propose(const double* x, const double* grad, const double* trust_radius, const uint8_t* active,
                        const double* cons_grads, const double* cons_values, const gabo_spd_acq_params* acq, void* workspace,
                        size_t workspace_bytes, double* x_prop, int64_t r, int d, int n_constraints, int n_equalities, double delta_cons,
                        double theta, double kappa, int mininner, int maxinner, int* any_active, int* status, gabo_stream_t stream) {
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return GABO_ERR_DIM;
    if (r < 0 || r > 0x7fffffffLL || n_constraints < 0 || n_constraints > gabo::kMaxCons || n_equalities < 0 ||
        n_equalities > n_constraints || maxinner < 1 || !acq)
        return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !grad || !trust_radius || !active || !workspace || !x_prop || !any_active || !status ||
        (n_constraints > 0 && (!cons_grads || !cons_values)))
        return GABO_ERR_ARG;
    if (acq->n < 1 || acq->n > gabo_spd_acq_max_train(d) || !acq->train_factors || !acq->alpha) return GABO_ERR_ARG;
    if (acq->kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!acq->linv || !acq->linv_t)) return GABO_ERR_ARG;
    {
        const int metric = acq->flags & GABO_METRIC_MASK;
        if (metric != GABO_METRIC_AFFINE_INVARIANT && metric != GABO_METRIC_LOG_EUCLIDEAN && metric != GABO_METRIC_FROBENIUS) return GABO_ERR_ARG;
        if (!gabo_spd_tr_propose_supported(acq->flags, d)) return GABO_ERR_DIM;
    }
    if (workspace_bytes < gabo_spd_tr_workspace_bytes(r, d, n_constraints, acq->n)) return GABO_ERR_ARG;
    gabo::ProposeArgs a{x, grad, trust_radius, active, cons_grads, cons_values, acq, workspace, x_prop, r, d, n_constraints, n_equalities,
                        delta_cons, theta, kappa, mininner, maxinner, any_active, status, (hipStream_t)stream};
    switch (acq->flags & GABO_METRIC_MASK) {
        case GABO_METRIC_AFFINE_INVARIANT: return gabo::propose_affine_invariant(a);
        case GABO_METRIC_LOG_EUCLIDEAN: return gabo::propose_log_euclidean(a);
        case GABO_METRIC_FROBENIUS: return gabo::propose_frobenius(a);
    }
    return GABO_ERR_DIM;
}

int gabo_spd_tr_update(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                       const uint8_t* invalid, const double* x_prop, void* workspace, int64_t r, int d, int n_constraints,
                       int64_t n_train, double delta_bar, double rho_prime, double rho_regularization, double mingradnorm,
                       int64_t maxiter, int* any_active, gabo_stream_t stream) {
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return GABO_ERR_DIM;
    if (r < 0 || r > 0x7fffffffLL || n_constraints < 0 || n_constraints > gabo::kMaxCons || n_train < 1) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !fx || !grad || !grad_norm || !trust_radius || !active || !iters || !x_prop || !workspace || !any_active)
        return GABO_ERR_ARG;
    size_t lds = (size_t)(4 * d * d) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_tr_update_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, x, fx, grad, grad_norm,
                       trust_radius, active, iters, invalid, x_prop, workspace, r, d, n_constraints, n_train, delta_bar, rho_prime,
                       rho_regularization, mingradnorm, maxiter, any_active);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_tr_solve_supported(const gabo_spd_acq_params* acq, int64_t r, int d, int n_constraints, int lift_dim) {
    if (!acq || d < 2 || d > 8 || r < 1 || n_constraints < 0 || n_constraints > gabo::kMaxCons) return 0;
    const int metric = acq->flags & GABO_METRIC_MASK;
    if (metric != GABO_METRIC_AFFINE_INVARIANT && metric != GABO_METRIC_LOG_EUCLIDEAN && metric != GABO_METRIC_FROBENIUS) return 0;
    if (acq->n < 1 || acq->n > gabo_spd_acq_max_train(d)) return 0;
    const size_t nested_bytes = lift_dim > 0 ? gabo::nested_extremes_lds_doubles(lift_dim, d) * sizeof(double) : 0;
#ifdef GABO_TR_NO_LAT
    const bool has_factors = false;
#else
    const bool has_factors = acq->linv && acq->linv_t;
#endif
    return gabo::solve_supported(metric == GABO_METRIC_LOG_EUCLIDEAN ? 1 : metric == GABO_METRIC_FROBENIUS ? 2 : 0, acq->n, r, d, n_constraints, has_factors, nested_bytes,
                                 gabo::tr_factor_count(*acq)) ? 1 : 0;
}

int gabo_tr_solve_record(double* buffer, int64_t max_iterations) {
    if (max_iterations < 0 || (max_iterations > 0 && !buffer)) return GABO_ERR_ARG;
    gabo::g_tr_record = max_iterations > 0 ? buffer : nullptr;
    gabo::g_tr_record_cap = max_iterations;
    return GABO_OK;
}

int gabo_spd_tr_solve(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                      const gabo_spd_acq_params* acq, int n_constraints, const int* constraint_kind, const double* constraint_bound,
                      int strict, void* workspace, size_t workspace_bytes, int64_t r, int d, double delta_cons, double theta, double kappa,
                      int mininner, int maxinner, double delta_bar, double rho_prime, double rho_regularization, double mingradnorm,
                      int64_t maxiter, const double* lift_w, const double* lift_p, const double* lift_x0, int lift_dim, int* status,
                      gabo_stream_t stream) {
    // (a pending record buffer belongs to THIS call whether it launches or not: taken before any early return, so that a call refused for its
    // arguments cannot leave the pointer to a later, unrelated solve)
    double* rec = nullptr;
    int64_t rec_cap = 0;
    gabo::tr_record_take(&rec, &rec_cap);
    if (d < 2 || d > 8) return GABO_ERR_DIM;
    if (r < 0 || r > 0x7fffffffLL || n_constraints < 0 || n_constraints > gabo::kMaxCons || maxinner < 1 || maxiter < 1 || !acq)
        return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !fx || !grad || !grad_norm || !trust_radius || !active || !iters || !workspace || !status ||
        (n_constraints > 0 && (!constraint_kind || !constraint_bound)))
        return GABO_ERR_ARG;
    if (acq->n < 1 || acq->n > gabo_spd_acq_max_train(d) || !acq->train_factors || !acq->alpha) return GABO_ERR_ARG;
    if (acq->kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!acq->linv || !acq->linv_t)) return GABO_ERR_ARG;
    if (workspace_bytes < gabo_spd_tr_workspace_bytes(r, d, n_constraints, acq->n)) return GABO_ERR_ARG;
    gabo::BuiltinCons B;
    B.n = n_constraints;
    B.strict = strict ? 1 : 0;
    B.big_dim = lift_dim;
    B.lift_w = lift_w;
    B.lift_p = lift_p;
    B.lift_x0 = lift_x0;
    for (int k = 0; k < gabo::kMaxCons; ++k) {
        B.kind[k] = k < n_constraints ? constraint_kind[k] : 0;
        B.bound[k] = k < n_constraints ? constraint_bound[k] : 0.0;
        if (k < n_constraints && (B.kind[k] < GABO_CONSTRAINT_MAX_EIGENVALUE || B.kind[k] > GABO_CONSTRAINT_MIN_EIGENVALUE_NESTED)) return GABO_ERR_ARG;
    }
    if (gabo::builtin_has_kind(B, true)) {
        // (the lifted dimension goes through the wave eigen-solver: kWaveEighMinDim <= lift_dim)
        if (!lift_w || !lift_p || !lift_x0) return GABO_ERR_ARG;
        if (lift_dim <= d || lift_dim < 5 || lift_dim > gabo::kTrNestedMaxDim) return GABO_ERR_DIM;
    }
    gabo::SolveArgs a{x, fx, grad, grad_norm, trust_radius, active, iters, acq, B, workspace, r, d, delta_cons, theta, kappa, mininner,
                      maxinner, delta_bar, rho_prime, rho_regularization, mingradnorm, maxiter, status, (hipStream_t)stream};
    a.rec = rec;
    a.rec_cap = rec_cap;
    return gabo::tr_solve_dispatch(a);
}

}  // extern "C"
