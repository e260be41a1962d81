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
#include "spd_tr_body.hpp"

namespace gabo {

// rho test and state update (robust_trust_regions.py:236-330; same algebra as batched_trust_regions.BatchedTrustRegions.solve)
__global__ __launch_bounds__(64) void spd_tr_update_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                           double* __restrict__ ng, double* __restrict__ delta_tr,
                                                           uint8_t* __restrict__ active, int64_t* __restrict__ iters,
                                                           const uint8_t* __restrict__ invalid, const double* __restrict__ x_prop,
                                                           void* wsbase, int64_t R, int d, int C, int64_t n, double delta_bar,
                                                           double rho_prime, double rho_regularization, double mingradnorm,
                                                           int64_t maxiter, int* __restrict__ any_active) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    const int64_t i = blockIdx.x;
    if (active[i] == 0) return;
    TrWs t = tr_layout(wsbase, R, d, C, n);
    const bool inval = invalid != nullptr && invalid[i] != 0;
    const double fx0 = fx[i];
    const double fxp = inval ? __builtin_inf() : t.fx_prop[i];
    const double rho_reg = (__builtin_fabs(fx0) > 1.0 ? __builtin_fabs(fx0) : 1.0) * 2.220446049250313e-16 * rho_regularization;
    const double rhonum = (fx0 - fxp) + rho_reg;
    const double rhoden = t.rhoden[i] + rho_reg;
    const bool model_decreased = rhoden >= 0.0;
    const double rho = rhoden == 0.0 ? __builtin_nan("") : rhonum / rhoden;
    const bool shrink = (rho < 0.25) || !model_decreased || (rho != rho) || inval;
    const int stop_inner = t.tcg.stop[i];
    const bool boundary = stop_inner == TCG_NEGATIVE_CURVATURE || stop_inner == TCG_EXCEEDED_TR ||
                          (C > 0 && stop_inner == TCG_REACHED_CONSTRAINTS);
    const bool grow = !shrink && rho > 0.75 && boundary;
    const double D0 = delta_tr[i];
    const double Dn = shrink ? D0 / 4 : (grow ? (2 * D0 < delta_bar ? 2 * D0 : delta_bar) : D0);
    const bool accept = model_decreased && rho > rho_prime;
    double ngi = ng[i];
    if (accept) {
        // x <- x+, g <- x+ sym(egrad) x+ (egrad2rgrad), ||g||_x = sqrt(tr(S x S x))
        lds_load(x_prop + i * dd, M0, d);
        lds_from_mandel(t.eg_prop + i * (int64_t)(d * (d + 1) / 2), M1, d);
        lds_mm(M1, M0, M2, d, false, false);          // P = S X
        lds_mm(M0, M2, M3, d, false, false);          // X S X
        double s = 0.0;
        for (int e = threadIdx.x; e < dd; e += 64) {
            int r = e / d, c = e - r * d;
            s = __builtin_fma(M2[e], M2[c * d + r], s);
            x[i * dd + e] = M0[e];
            g[i * dd + e] = 0.5 * (M3[e] + M3[c * d + r]);
        }
        s = wave_sum(s);
        ngi = __builtin_sqrt(s > 0.0 ? s : 0.0);
    }
    if (threadIdx.x == 0) {
        delta_tr[i] = Dn;
        if (accept) { fx[i] = fxp; ng[i] = ngi; }
        const int64_t it = iters[i] + 1;
        iters[i] = it;
        if (ngi < mingradnorm || it >= maxiter) active[i] = 0;
        else atomicOr(any_active, 1);
    }
}

int propose_affine_invariant(const ProposeArgs& a) { return dispatch_propose<0, 12>(a); }

}  // namespace gabo

extern "C" {

size_t gabo_spd_tr_workspace_bytes(int64_t r, int d, int n_constraints, int64_t n_train) {
    if (r < 0 || d < 1 || n_constraints < 0 || n_train < 0) return 0;
    return gabo::tr_layout(nullptr, r, d, n_constraints, n_train).bytes;
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
        if (metric != GABO_METRIC_AFFINE_INVARIANT && metric != GABO_METRIC_LOG_EUCLIDEAN && metric != GABO_METRIC_FROBENIUS)
            return GABO_ERR_ARG;
        if (metric != GABO_METRIC_AFFINE_INVARIANT && d > 8) return GABO_ERR_DIM;
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

}  // extern "C"
