// SPD affine-invariant pairwise kernel and its gradient for 12 < d <= 32: ONE WAVE PER PAIR, d x d tiles staged in LDS,
// cyclic Jacobi with the rotation applied in parallel over the row/column index.  The register-resident lane-per-pair
// kernels (spd_pairwise.hip / spd_backward.hip) stop at d = 12 because M no longer fits the VGPR file; this fallback keeps
// the whole range of the C ABI served.  It is a correctness path (a few 1e5 pairs/s), not the metric.
#include "gabo_device.hpp"
#include "lds_linalg.hpp"
#include "spd_generic.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

// W = L^-1 (full d x d, row major) for every x1 matrix: ws[(b*n1 + i) * d*d]
__global__ __launch_bounds__(64) void spd_prep_generic_kernel(const double* __restrict__ x, double* __restrict__ ws, int64_t n,
                                                              int64_t batch_stride, int d, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* A = lds;
    double* W = lds + d * d;
    const int64_t g = blockIdx.x;
    const int64_t b = g / n, i = g - b * n;
    const int dv = d * (d + 1) / 2;
    lds_from_mandel(x + b * batch_stride + i * dv, A, d);
    bool ok = lds_cholesky(A, d);
    lds_tri_inverse(A, W, d);
    lds_store(W, ws + g * d * d, d);
    if (!ok && threadIdx.x == 0 && atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)g;
}

// spd2 check of the second set: a non-SPD x2 matrix is flagged too (the reference would fail later with NaNs)
__device__ __forceinline__ double pair_weight(double d2, double go, double beta, int mode) {
    double dist = __builtin_sqrt(d2);
    if (mode == GABO_OUT_GAUSSIAN) return go * (-beta) * exp(-((dist * dist) * beta));
    if (mode == GABO_OUT_LAPLACE) return go * (-beta) * exp(-(dist * beta)) / (2.0 * dist);
    return go / (2.0 * dist);
}

// block = one pair (b, i, j).  BWD = false: out/dist_out.  BWD = true: atomically accumulates w_ij logm(M_ij) into S[b,i].
template <bool BWD, bool QL>
__global__ __launch_bounds__(64) void spd_pair_generic_kernel(const double* __restrict__ Wg, const double* __restrict__ x2,
                                                              double* __restrict__ out, double* __restrict__ dist_out,
                                                              const double* __restrict__ gout, double* __restrict__ S, int64_t n1,
                                                              int64_t n2, int d, int64_t w_bs, int64_t x2_bs, int64_t go_sb,
                                                              int64_t go_si, int64_t go_sj, double beta, int flags) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* Wl = lds;
    double* B = Wl + dd;
    double* T = B + dd;
    double* M = T + dd;
    double* V = M + dd;      // BWD only
    double* cs = V + dd;
    const int mode = flags & GABO_OUT_MASK;
    const int64_t g = blockIdx.x;
    const int64_t b = g / (n1 * n2);
    const int64_t rem = g - b * n1 * n2;
    const int64_t i = rem / n2, j = rem - i * n2;
    const int dv = d * (d + 1) / 2;
    lds_load(Wg + b * w_bs + i * dd, Wl, d);
    lds_from_mandel(x2 + b * x2_bs + j * dv, B, d);
    lds_congruence(Wl, B, M, T, d);
    lds_symmetrize(M, T, d);
    lds_eigh<QL>(M, BWD ? V : nullptr, cs, d);
    double s = 0.0;
    for (int k = 0; k < d; ++k) { double lg = log(M[k * d + k]); s = __builtin_fma(lg, lg, s); }
    const double d2 = s + 1e-15;
    if constexpr (!BWD) {
        if (threadIdx.x == 0) {
            double dist = __builtin_sqrt(d2);
            double val = mode == GABO_OUT_DISTANCE ? dist : (mode == GABO_OUT_LAPLACE ? exp(-(dist * beta)) : exp(-((dist * dist) * beta)));
            out[g] = val;
            if (dist_out) dist_out[g] = dist;
        }
    } else {
        const double w = pair_weight(d2, gout[b * go_sb + i * go_si + j * go_sj], beta, mode);
        double* Si = S + (b * n1 + i) * dd;
        for (int e = threadIdx.x; e < dd; e += blockDim.x) {
            int r = e / d, c = e - r * d;
            double f = 0.0;
            for (int k = 0; k < d; ++k) f = __builtin_fma(V[r * d + k] * log(M[k * d + k]), V[c * d + k], f);
            unsafeAtomicAdd(Si + e, w * f);      // fp64 hardware atomic; summation order over j is not fixed
        }
    }
}

// grad_A = -2 W^T S W, then Mandel.  block per (b, i)
__global__ __launch_bounds__(64) void spd_bwd_finalize_generic_kernel(const double* __restrict__ Wg, const double* __restrict__ S,
                                                                      double* __restrict__ gx, int d, int64_t w_bs, int64_t n1) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* Wl = lds;
    double* Sl = Wl + dd;
    double* T = Sl + dd;
    double* Gm = T + dd;
    const int64_t g = blockIdx.x;
    const int64_t b = g / n1, i = g - b * n1;
    lds_load(Wg + b * w_bs + i * dd, Wl, d);
    lds_load(S + g * dd, Sl, d);
    lds_mm(Wl, Sl, T, d, true, false);     // W^T S
    lds_mm(T, Wl, Gm, d, false, false);    // W^T S W
    const int dv = d * (d + 1) / 2;
    for (int e = threadIdx.x; e < dv; e += blockDim.x) {
        int k = 0;
        while (k + 1 < d && (k + 1) * d - (k + 1) * k / 2 <= e) ++k;
        int c = e - (k * d - k * (k - 1) / 2);
        int r = c + k;
        double v = -2.0 * 0.5 * (Gm[r * d + c] + Gm[c * d + r]);
        gx[g * dv + e] = (k == 0) ? v : v * kSqrt2;
    }
}

int launch_spd_ai_generic(const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                          int d, int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st) {
    const int64_t b1 = (s1 == 0) ? 1 : batch;
    const int64_t dd = (int64_t)d * d;
    if (b1 * n1 > 0x7fffffffLL || batch * n1 * n2 > 0x7fffffffLL) return GABO_ERR_ARG;
    hipLaunchKernelGGL(spd_prep_generic_kernel, dim3((unsigned)(b1 * n1)), dim3(64), (size_t)(2 * dd) * 8, st, x1, ws, n1, s1, d, status);
#define GABO_GENERIC_FWD(QL)                                                                                                          \
    hipLaunchKernelGGL((spd_pair_generic_kernel<false, QL>), dim3((unsigned)(batch * n1 * n2)), dim3(64), (size_t)(5 * dd + kJacobiScratch) * 8, \
                       st, ws, x2, out, dist_out, (const double*)nullptr, (double*)nullptr, n1, n2, d, (s1 == 0) ? (int64_t)0 : n1 * dd, \
                       s2, (int64_t)0, (int64_t)0, (int64_t)0, beta, flags & ~GABO_SYMMETRIC)
    if (d >= kWaveEighMinDim) GABO_GENERIC_FWD(true);
    else GABO_GENERIC_FWD(false);
#undef GABO_GENERIC_FWD
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int launch_spd_ai_backward_generic(const double* x1, const double* x2, const double* gout, double* gx, int64_t batch, int64_t n1,
                                   int64_t n2, int d, int64_t s1, int64_t s2, int64_t go_sb, int64_t go_si, int64_t go_sj,
                                   double beta, int flags, double* ws, int* status, hipStream_t st) {
    const int64_t b1 = (s1 == 0) ? 1 : batch;
    const int64_t dd = (int64_t)d * d;
    if (batch * n1 > 0x7fffffffLL || batch * n1 * n2 > 0x7fffffffLL) return GABO_ERR_ARG;
    double* W = ws;
    double* S = ws + b1 * n1 * dd;
    hipLaunchKernelGGL(spd_prep_generic_kernel, dim3((unsigned)(b1 * n1)), dim3(64), (size_t)(2 * dd) * 8, st, x1, W, n1, s1, d, status);
    hipMemsetAsync(S, 0, (size_t)(batch * n1 * dd) * 8, st);
#define GABO_GENERIC_BWD(QL)                                                                                                          \
    hipLaunchKernelGGL((spd_pair_generic_kernel<true, QL>), dim3((unsigned)(batch * n1 * n2)), dim3(64), (size_t)(5 * dd + kJacobiScratch) * 8, \
                       st, W, x2, (double*)nullptr, (double*)nullptr, gout, S, n1, n2, d, (s1 == 0) ? (int64_t)0 : n1 * dd, s2, go_sb,    \
                       go_si, go_sj, beta, flags)
    if (n2 > 0) {
        if (d >= kWaveEighMinDim) GABO_GENERIC_BWD(true);
        else GABO_GENERIC_BWD(false);
    }
#undef GABO_GENERIC_BWD
    hipLaunchKernelGGL(spd_bwd_finalize_generic_kernel, dim3((unsigned)(batch * n1)), dim3(64), (size_t)(4 * dd) * 8, st, W, S, gx, d,
                       (s1 == 0) ? (int64_t)0 : n1 * dd, n1);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // namespace gabo
