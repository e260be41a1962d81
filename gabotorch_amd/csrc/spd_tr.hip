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
#include "spd_acq_body.hpp"
#include "spd_tcg_body.hpp"

namespace gabo {

struct TrWs {
    TcgWs tcg;
    double *x_fd, *eg_fd, *val_fd, *xp_mandel, *eg_prop, *fx_prop, *rhoden, *F;
    size_t bytes;
};

static __host__ __device__ inline TrWs tr_layout(void* base, int64_t R, int d, int C, int64_t n) {
    TrWs t;
    t.tcg = tcg_layout(base, R, d, C);
    size_t off = (t.tcg.bytes + 15) & ~(size_t)15;
    double* p = (double*)((char*)base + off);
    const int64_t dv = (int64_t)d * (d + 1) / 2;
    t.x_fd = p;       p += R * dv;
    t.eg_fd = p;      p += R * dv;
    t.val_fd = p;     p += R;
    t.xp_mandel = p;  p += R * dv;
    t.eg_prop = p;    p += R * dv;
    t.fx_prop = p;    p += R;
    t.rhoden = p;     p += R;
    t.F = p;          p += R * dv * n;
    t.bytes = (size_t)((char*)p - (char*)base);
    return t;
}

template <int D>
__global__ __launch_bounds__(64) void spd_tr_propose_kernel(const double* __restrict__ x, const double* __restrict__ g,
                                                            const double* __restrict__ delta_tr, const uint8_t* __restrict__ active,
                                                            const double* __restrict__ gc, const double* __restrict__ fc,
                                                            AcqParams P, void* wsbase, double* __restrict__ x_prop, int64_t R, int C,
                                                            int neq, double delta_cons, double theta, double kappa, int mininner,
                                                            int maxinner, int* __restrict__ any_active, int* __restrict__ status) {
    constexpr int T = tri_size(D);
    constexpr int dd = D * D;
    __shared__ AcqLds<D> acq;
    __shared__ __attribute__((aligned(16))) double mats[5 * dd + 2];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    if (i == 0 && threadIdx.x == 0) *any_active = 0;          // set again by the update kernel
    if (active[i] == 0) return;
    TrWs t = tr_layout(wsbase, R, D, C, P.n);
    const TcgWs& w = t.tcg;
    double* xfd = t.x_fd + i * T;
    double* egfd = t.eg_fd + i * T;
    double* F = t.F + i * T * P.n;
    tcg_begin(x + i * dd, g + i * dd, gc, fc, true, delta_tr[i], w, i, R, D, C, status, mats);
    __syncthreads();
    for (int it = 0; it < maxinner; ++it) {
        tcg_fd_point(w, i, D, xfd, mats);
        __syncthreads();
        acq_eval<D>(xfd, P, t.val_fd + i, egfd, F, acq, dyn, status, i);
        __syncthreads();
        const bool running = tcg_step(w, i, R, D, C, egfd, neq, delta_cons, theta, kappa, mininner, it, mats);
        __syncthreads();
        if (!running) break;
    }
    // ---- proposal x+ = L expm(eta~) L^T and the model decrease -<g, eta> - 1/2 <eta, H eta> (whitened Frobenius dots)
    double* M0 = mats;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* cs = mats + 5 * dd;
    const double* etaw = w.eta_w + i * dd;
    const double ge = wave_dot(w.g_w + i * dd, etaw, dd);
    const double ehe = wave_dot(etaw, w.heta_w + i * dd, dd);
    if (threadIdx.x == 0) t.rhoden[i] = -ge - 0.5 * ehe;
    lds_load(w.chol + i * dd, M0, D);
    if constexpr (D <= 8) {
        // expm(eta~) by the register Jacobi (every lane redundantly, no barriers); lane 0 publishes E
        double m[T], v[D * D];
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            static_for<r + 1>([&](auto cc) { constexpr int c = decltype(cc)::value; m[tri(r, c)] = 0.5 * (etaw[r * D + c] + etaw[c * D + r]); });
        });
        jacobi_eig_reg<D>(m, v);
        double ex[D];
        static_for<D>([&](auto kk) { ex[decltype(kk)::value] = exp(m[tri(decltype(kk)::value, decltype(kk)::value)]); });
        if (threadIdx.x == 0) {
            static_for<D>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<r + 1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    double f = 0.0;
                    static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; f = __builtin_fma(v[r * D + k] * ex[k], v[c * D + k], f); });
                    M3[r * D + c] = f;
                    M3[c * D + r] = f;
                });
            });
        }
        __syncthreads();
    } else {
        lds_load(etaw, M1, D);
        lds_jacobi(M1, M2, cs, D);
        lds_fun_from_eig(M1, M2, M3, D, FN_EXP);
    }
    lds_congruence(M0, M3, M1, M2, D);
    lds_symmetrize(M1, M2, D);
    lds_store(M1, x_prop + i * dd, D);
    double* xpm = t.xp_mandel + i * T;
    for (int e = threadIdx.x; e < T; e += 64) {
        int k = 0;
        while (k + 1 < D && (k + 1) * D - (k + 1) * k / 2 <= e) ++k;
        int cc = e - (k * D - k * (k - 1) / 2);
        int r = cc + k;
        xpm[e] = (k == 0) ? M1[r * D + cc] : kSqrt2 * M1[r * D + cc];
    }
    __syncthreads();
    acq_eval<D>(xpm, P, t.fx_prop + i, t.eg_prop + i * T, F, acq, dyn, status, i);
}

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

template <int D>
static int launch_propose(const double* x, const double* g, const double* delta_tr, const uint8_t* active, const double* gc,
                          const double* fc, const AcqParams& P, void* ws, double* x_prop, int64_t r, int c, int neq, double delta_cons,
                          double theta, double kappa, int mininner, int maxinner, int* any_active, int* status, hipStream_t st) {
    size_t lds = (size_t)(3 * P.n) * sizeof(double);
    hipLaunchKernelGGL((spd_tr_propose_kernel<D>), dim3((unsigned)r), dim3(64), lds, st, x, g, delta_tr, active, gc, fc, P, ws, x_prop,
                       r, c, neq, delta_cons, theta, kappa, mininner, maxinner, any_active, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

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
    if (workspace_bytes < gabo_spd_tr_workspace_bytes(r, d, n_constraints, acq->n)) return GABO_ERR_ARG;
#define GABO_CASE(DD) \
    case DD:          \
        return gabo::launch_propose<DD>(x, grad, trust_radius, active, cons_grads, cons_values, *acq, workspace, x_prop, r,       \
                                        n_constraints, n_equalities, delta_cons, theta, kappa, mininner, maxinner, any_active, \
                                        status, (hipStream_t)stream);
    switch (d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
    }
#undef GABO_CASE
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
