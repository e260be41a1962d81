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
#include "gabo_device.hpp"
#include "spd_prep.hpp"
#include "spd_jacobi.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

static __device__ __forceinline__ double wave_sum64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int D>
__global__ __launch_bounds__(64) void spd_acq_kernel(const double* __restrict__ x, const double* __restrict__ G,
                                                     const double* __restrict__ alpha, const double* __restrict__ linv,
                                                     const double* __restrict__ linv_t, double* __restrict__ value,
                                                     double* __restrict__ grad, double* __restrict__ scratch, int64_t n, double beta,
                                                     int mode, double mean0, double os, double kxx, double best_f, int kind,
                                                     int maximize, double out_sign, const int* __restrict__ active, int* __restrict__ status) {
    constexpr int T = tri_size(D);
    constexpr int LD = 64;
    if (active && active[blockIdx.x] == 0) return;     // masked candidate: outputs left untouched
    __shared__ double acc[T * LD];
    __shared__ double vls[D * D * 64];
    __shared__ double red[T];
    __shared__ double wl[T];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    double* ks = dyn;          // n : outputscale * k_j
    double* kd = ks + n;       // n : d k_j / d (d_j^2)
    double* vv = kd + n;       // n : L^-1 ks
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    const bool want_grad = grad != nullptr;
    // ---- candidate side: W = chol(x*)^-1, computed by every lane (wave-uniform, a few hundred flops)
    // and kept in LDS (broadcast reads with compile-time offsets) so that it does not compete with M for registers
    {
        double a[T], w[T];
        const bool bad = mandel_cholesky<D>(x + i * T, a);
        if (bad && lane == 0 && status) {
            if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)i;
        }
        lower_inverse<D>(a, w);
        if (lane == 0) static_for<T>([&](auto ee) { wl[decltype(ee)::value] = w[decltype(ee)::value]; });
    }
    __syncthreads();
    const double* w = wl;
    double* F = scratch + i * T * n;      // logm(M_j), lower triangle, entry-major [T][n]
    for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool live = j < n;
        const double* Gj = G + (live ? j : n - 1);
        double m[T];
        static_for<T>([&](auto ee) { m[decltype(ee)::value] = 0.0; });
        static_for<D>([&](auto cc) {
            constexpr int col = decltype(cc)::value;
            double g[D - col], c[D - col];
            static_for<D - col>([&](auto kk) { g[decltype(kk)::value] = Gj[(int64_t)tri(col + decltype(kk)::value, col) * n]; });
            static_for<D - col>([&](auto rr) {
                constexpr int r = col + decltype(rr)::value;
                double a = w[tri(r, col)] * g[0];
                static_for<r - col>([&](auto kk) {
                    constexpr int k = col + 1 + decltype(kk)::value;
                    a = __builtin_fma(w[tri(r, k)], g[k - col], a);
                });
                c[r - col] = a;
            });
            static_for<D - col>([&](auto rr) {
                constexpr int r = col + decltype(rr)::value;
                static_for<r - col + 1>([&](auto qq) {
                    constexpr int q = col + decltype(qq)::value;
                    m[tri(r, q)] = __builtin_fma(c[r - col], c[q - col], m[tri(r, q)]);
                });
            });
        });
        double* vl = vls + lane;
        jacobi_eig<D>(m, vl);
        double lg[D];
        double s = 0.0;
        static_for<D>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            lg[k] = log(m[tri(k, k)]);
            s = __builtin_fma(lg[k], lg[k], s);
        });
        const double d2 = s + 1e-15;                      // spd_utils_torch.py:120
        const double dist = __builtin_sqrt(d2);
        double kj, dk;
        if (mode == GABO_OUT_GAUSSIAN) {
            kj = exp(-((dist * dist) * beta));
            dk = -beta * kj;
        } else {
            kj = exp(-(dist * beta));
            dk = -beta * kj / (2.0 * dist);
        }
        if (live) {
            ks[j] = os * kj;
            kd[j] = dk;
            if (want_grad) {
                static_for<D>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    static_for<r + 1>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        double f = 0.0;
                        static_for<D>([&](auto kk) {
                            constexpr int k = decltype(kk)::value;
                            f = __builtin_fma(vl[(r * D + k) * 64] * lg[k], vl[(c * D + k) * 64], f);
                        });
                        F[(int64_t)tri(r, c) * n + j] = f;
                    });
                });
            }
        }
    }
    __syncthreads();
    // ---- exact-GP posterior + acquisition (same arithmetic as gp_acquisition.hip)
    double part = 0.0;
    for (int64_t j = lane; j < n; j += 64) part = __builtin_fma(ks[j], alpha[j], part);
    const double mean = mean0 + wave_sum64(part);
    const double sgn = maximize ? 1.0 : -1.0;
    double g_mean, g_var = 0.0;
    if (kind == GABO_ACQ_POSTERIOR_MEAN) {
        if (lane == 0) value[i] = out_sign * sgn * mean;
        g_mean = sgn;
    } else {
        part = 0.0;
        for (int64_t r = lane; r < n; r += 64) {
            double a = 0.0;
            for (int64_t j = 0; j <= r; ++j) a = __builtin_fma(linv_t[j * n + r], ks[j], a);
            vv[r] = a;
            part = __builtin_fma(a, a, part);
        }
        const double var = os * kxx - wave_sum64(part);
        const bool clamped = !(var > 1e-9);
        const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
        const double u = sgn * (mean - best_f) / sigma;
        const double pdf = exp(-0.5 * u * u) * 0.3989422804014327;
        const double cdf = 0.5 * (1.0 + erf(u * 0.7071067811865476));
        if (lane == 0) value[i] = out_sign * sigma * (pdf + u * cdf);
        g_mean = sgn * cdf;
        g_var = clamped ? 0.0 : 0.5 * pdf / sigma;
    }
    if (!want_grad) return;
    __syncthreads();
    // ---- weights w_j = d(out_sign acq)/d(d_j^2) and S = sum_j w_j logm(M_j), one accumulator column per lane
    static_for<T>([&](auto ee) { acc[decltype(ee)::value * LD + lane] = 0.0; });
    for (int64_t j = lane; j < n; j += 64) {
        double ws = 0.0;
        if (kind != GABO_ACQ_POSTERIOR_MEAN)
            for (int64_t r = j; r < n; ++r) ws = __builtin_fma(linv[r * n + j], vv[r], ws);
        const double gk = out_sign * os * (g_mean * alpha[j] - 2.0 * g_var * ws);
        const double wj = gk * kd[j];
        static_for<T>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            acc[e * LD + lane] = __builtin_fma(wj, F[(int64_t)e * n + j], acc[e * LD + lane]);
        });
    }
    __syncthreads();
    for (int e = lane; e < T; e += 64) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += acc[e * LD + ((l + e) & 63)];
        red[e] = t;
    }
    __syncthreads();
    // grad = -2 W^T S W, Mandel  (as in spd_backward.hip)
    for (int e = lane; e < T; e += 64) {
        int a = 0;
        while (tri(a + 1, 0) <= e) ++a;
        int bb = e - tri(a, 0);
        double t = 0.0;
        for (int r = a; r < D; ++r) {
            double inner = 0.0;
            for (int c = bb; c < D; ++c) {
                double srs = r >= c ? red[tri(r, c)] : red[tri(c, r)];
                inner = __builtin_fma(srs, wl[tri(c, bb)], inner);
            }
            t = __builtin_fma(wl[tri(r, a)], inner, t);
        }
        t *= -2.0;
        grad[i * T + mandel_pos(D, a, bb)] = (a == bb) ? t : t * kSqrt2;
    }
}

template <int D>
static int launch_spd_acq(const double* x, const double* G, const double* alpha, const double* linv, const double* linv_t,
                          double* value, double* grad, double* scratch, int64_t r, int64_t n, double beta, int mode, double mean,
                          double os, double kxx, double best_f, int kind, int maximize, double out_sign, const int* active, int* status, hipStream_t st) {
    size_t lds = (size_t)(3 * n) * sizeof(double);
    hipLaunchKernelGGL((spd_acq_kernel<D>), dim3((unsigned)r), dim3(64), lds, st, x, G, alpha, linv, linv_t, value, grad, scratch, n,
                       beta, mode, mean, os, kxx, best_f, kind, maximize, out_sign, active, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

template <int D>
static int launch_prepare_train(const double* x, double* G, int64_t n, int* status, hipStream_t st) {
    hipLaunchKernelGGL((spd_prep_kernel<D>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, x, G, (int64_t)1, n, (int64_t)0, 1,
                       status, 0);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // namespace gabo

extern "C" {

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
    if (r < 0 || r > 0x7fffffffLL || n < 1 || n > 2048) return GABO_ERR_ARG;
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return GABO_ERR_DIM;
    if (flags != GABO_OUT_GAUSSIAN && flags != GABO_OUT_LAPLACE) return GABO_ERR_ARG;
    if (kind != GABO_ACQ_EXPECTED_IMPROVEMENT && kind != GABO_ACQ_POSTERIOR_MEAN) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x_mandel || !train_factors || !alpha || !value || !status || (grad_mandel && !scratch)) return GABO_ERR_ARG;
    if (kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!linv || !linv_t)) return GABO_ERR_ARG;
#define GABO_CASE(DD) \
    case DD:          \
        return gabo::launch_spd_acq<DD>(x_mandel, train_factors, alpha, linv, linv_t, value, grad_mandel, scratch, r, n, beta, flags, \
                                        mean, outputscale, kxx, best_f, kind, maximize, out_sign, active, status, (hipStream_t)stream);
    switch (d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

}  // extern "C"
