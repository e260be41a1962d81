// Kernel templates of the trust-region iteration (see spd_tr.hip for the description).
#pragma once
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

template <int D, int METRIC>
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
        acq_eval_any<D, METRIC>(xfd, P, t.val_fd + i, egfd, F, acq, dyn, status, i);
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
    acq_eval_any<D, METRIC>(xpm, P, t.fx_prop + i, t.eg_prop + i * T, F, acq, dyn, status, i);
}

// one translation unit per metric (spd_tr.hip, spd_tr_le.hip, spd_tr_frob.hip) so that the instantiations compile in parallel
template <int D, int METRIC>
static int launch_propose_one(const double* x, const double* g, const double* delta_tr, const uint8_t* active, const double* gc,
                              const double* fc, const AcqParams& P, void* ws, double* x_prop, int64_t r, int c, int neq,
                              double delta_cons, double theta, double kappa, int mininner, int maxinner, int* any_active, int* status,
                              hipStream_t st) {
    size_t lds = (size_t)(3 * P.n) * sizeof(double);
    hipLaunchKernelGGL((spd_tr_propose_kernel<D, METRIC>), dim3((unsigned)r), dim3(64), lds, st, x, g, delta_tr, active, gc, fc, P, ws,
                       x_prop, r, c, neq, delta_cons, theta, kappa, mininner, maxinner, any_active, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

struct ProposeArgs {
    const double *x, *g, *delta_tr;
    const uint8_t* active;
    const double *gc, *fc;
    const AcqParams* P;
    void* ws;
    double* x_prop;
    int64_t r;
    int d, c, neq;
    double delta_cons, theta, kappa;
    int mininner, maxinner;
    int *any_active, *status;
    hipStream_t st;
};

template <int METRIC, int DMAX>
static int dispatch_propose(const ProposeArgs& a) {
#define GABO_CASE(DD)                                                                                                                  \
    case DD:                                                                                                                           \
        if constexpr (DD <= DMAX)                                                                                                      \
            return launch_propose_one<DD, METRIC>(a.x, a.g, a.delta_tr, a.active, a.gc, a.fc, *a.P, a.ws, a.x_prop, a.r, a.c, a.neq, \
                                                  a.delta_cons, a.theta, a.kappa, a.mininner, a.maxinner, a.any_active, a.status, a.st); \
        else                                                                                                                           \
            return GABO_ERR_DIM;
    switch (a.d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

// defined in spd_tr.hip / spd_tr_le.hip / spd_tr_frob.hip
int propose_affine_invariant(const ProposeArgs& a);
int propose_log_euclidean(const ProposeArgs& a);
int propose_frobenius(const ProposeArgs& a);

}  // namespace gabo
