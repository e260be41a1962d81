// Gram matrix of the nested SPD kernels in TWO launches (config 5: hd_gabo_spd, S^20_++ -> S^2_++):
//   NestedSpdAffineInvariantGaussianKernel.forward   kernel_utils/kernels_nested_spd.py:104-136   project -> affine-invariant Gaussian Gram
//   NestedSpdLogEuclideanGaussianKernel.forward      kernel_utils/kernels_nested_spd.py:191-246   project -> logm -> exp(-||.||_F^2 / l^2)
//   projection_from_spd_to_nested_spd                nested_mappings/nested_spd_utils.py:13-48     Y = W^T X W
// The separate-launch chain was project (one launch per point set) -> [Cholesky / inverse preparation | logm] -> Gram: 3-4 launches of which
// all but the last handle a few thousand tiny matrices (5-7 us each, launch-bound: profiles/r03_config5_kernel_stats.csv).  Here ONE launch
// projects both point sets (one wave per D x D matrix: a coalesced read of its Mandel vector against the dl_vec x D_vec operator in LDS) and
// finishes each d x d result in the lanes' registers - Cholesky factor (+ inverse for x1) in the layouts of spd_prep.hpp, or the Mandel vector of
// its matrix logarithm - straight into the workspace of the Gram launch (spd_ai_gauss2_kernel / spd_ai_pairwise_kernel, frobenius_pairwise_kernel).
// No gradient (the differentiable route stays gabo_spd_project / gabo_spd_logm_mandel / the pairwise kernels and their backward launches).
#include "gabo_device.hpp"
#include "spd_eigvec.hpp"
#include "spd_prep.hpp"
#include "spd_project_operator.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

int launch_spd_ai_prepared(int d, double* out, int64_t batch, int64_t n1, int64_t n2, bool shared1, bool shared2, double beta, int flags,
                           double* ws, hipStream_t st);      // spd_pairwise.hip

// matrices [0, m1) come from x1, [m1, m1 + m2) from x2 (m2 = 0 with `same`: x2 is x1, one projection serves both roles).
// LOGM = false: o1 = W (chol^-1, packed lower, row contiguous, m1 rows), o2 = G (chol, entry-major [b][T][n2]).
// LOGM = true:  o1 / o2 = Mandel vectors of logm(Y) of the two sets (o2 unused with `same`).
template <int DL, bool LOGM>
__global__ __launch_bounds__(256) void nested_spd_prepare_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                                 const double* __restrict__ w, double* __restrict__ o1, double* __restrict__ o2,
                                                                 int64_t m1, int64_t m2, int64_t n2, int D, int same, int* __restrict__ status) {
    constexpr int DV = DL * (DL + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Dv = D * (D + 1) / 2;
    double* Wl = lds;                 // D x DL
    double* P = Wl + D * DL;          // DV x Dv
    build_projection_operator<DL>(w, D, Wl, P);
    const int lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    for (int64_t i = (int64_t)blockIdx.x * waves + (threadIdx.x >> 6); i < m1 + m2; i += (int64_t)gridDim.x * waves) {
        const bool second = i >= m1;
        const int64_t k = second ? i - m1 : i;
        double y[DV];
        project_one<DL>((second ? x2 : x1) + k * Dv, P, Dv, y);
        if constexpr (LOGM) {
            double m[DV], lam[DL], v[DL * DL];
            static_for<DL>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<r + 1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const double e = y[mandel_pos(DL, r, c)];
                    m[tri(r, c)] = (r == c) ? e : e / kSqrt2;           // (the division of spd_utils_torch.py:186-187, as gabo_spd_logm_mandel does)
                });
            });
            sym_eig_reg<DL>(m, lam, v);
            double lg[DL];
            static_for<DL>([&](auto kk) { lg[decltype(kk)::value] = log(lam[decltype(kk)::value]); });     // (NaN for a non-positive eigenvalue, like the reference)
            double* dst = (second ? o2 : o1) + k * DV;
            static_for<DL>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<r + 1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    double f = 0.0;
                    static_for<DL>([&](auto kk) { constexpr int q = decltype(kk)::value; f = __builtin_fma(v[r * DL + q] * lg[q], v[c * DL + q], f); });
                    if (lane == 0) dst[mandel_pos(DL, r, c)] = (r == c) ? f : f * kSqrt2;
                });
            });
        } else {
            double a[DV];
            bool bad = mandel_cholesky<DL>(y, a);
            const bool as_first = !second, as_second = second || same;
            if (as_first) {
                if (bad && lane == 0 && atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)k;
                double wi[DV];
                lower_inverse<DL>(a, wi);
                if (lane == 0) static_for<DV>([&](auto ee) { o1[k * DV + decltype(ee)::value] = wi[decltype(ee)::value]; });
            }
            if (as_second) {
                // (x2 of the forward Gram: a matrix that is not positive definite but free of NaN gives a NaN column, NaN entries are reported:
                // spd_prep.hpp, `lenient2`)
                if (bad && !as_first) {
                    bool has_nan = false;
                    static_for<DV>([&](auto ee) { has_nan |= y[decltype(ee)::value] != y[decltype(ee)::value]; });
                    if (has_nan) {
                        if (lane == 0 && atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)(m1 + k);
                    } else {
                        static_for<DV>([&](auto ee) { a[decltype(ee)::value] = __builtin_nan(""); });
                    }
                }
                const int64_t b = k / n2, col = k - b * n2;
                if (lane == 0) static_for<DV>([&](auto ee) { o2[(b * DV + decltype(ee)::value) * n2 + col] = a[decltype(ee)::value]; });
            }
        }
    }
}

template <int DL, bool LOGM>
static void launch_nested_prepare(const double* x1, const double* x2, const double* w, double* o1, double* o2, int64_t m1, int64_t m2, int64_t n2,
                                  int D, int same, int* status, hipStream_t st) {
    const size_t lds = (size_t)(D * DL + (DL * (DL + 1) / 2) * (D * (D + 1) / 2)) * sizeof(double);
    const int64_t total = m1 + m2;
    const int64_t blocks = (total + 3) / 4 < 4096 ? (total + 3) / 4 : 4096;
    hipLaunchKernelGGL((nested_spd_prepare_kernel<DL, LOGM>), dim3((unsigned)blocks), dim3(256), lds, st, x1, x2, w, o1, o2, m1, m2, n2, D, same, status);
}

}  // namespace gabo

extern "C" size_t gabo_nested_spd_gram_workspace_bytes(int64_t batch, int64_t n1, int64_t n2, int dl) {
    if (batch <= 0 || n1 < 0 || n2 < 0 || dl < 2) return 16;
    return (size_t)(batch * (n1 + n2)) * (size_t)(dl * (dl + 1) / 2) * sizeof(double) + 16;
}

extern "C" int gabo_nested_spd_gram(const double* x1, const double* x2, const double* w, double* out, int64_t batch, int64_t n1, int64_t n2, int D,
                                    int dl, int metric, double beta, int flags, void* workspace, size_t workspace_bytes, int* status,
                                    gabo_stream_t stream) {
    if (batch < 0 || n1 < 0 || n2 < 0) return GABO_ERR_ARG;
    if (D < 2 || D > GABO_SPD_MAX_DIM || dl < 2 || dl > 4 || dl > D) return GABO_ERR_DIM;
    if (metric != GABO_METRIC_AFFINE_INVARIANT && metric != GABO_METRIC_LOG_EUCLIDEAN) return GABO_ERR_ARG;
    if (batch == 0 || n1 == 0 || n2 == 0) return GABO_OK;
    if (!x1 || !x2 || !w || !out || !workspace || !status) return GABO_ERR_ARG;
    if (workspace_bytes < gabo_nested_spd_gram_workspace_bytes(batch, n1, n2, dl)) return GABO_ERR_ARG;
    const size_t lds = (size_t)(D * dl + (dl * (dl + 1) / 2) * (D * (D + 1) / 2)) * sizeof(double);
    if (lds > 48 * 1024) return GABO_ERR_DIM;
    if ((flags & GABO_SYMMETRIC) && (n1 != n2 || x1 != x2)) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int T = dl * (dl + 1) / 2;
    const int same = (x1 == x2 && n1 == n2) ? 1 : 0;
    const int64_t m1 = batch * n1, m2 = same ? 0 : batch * n2;
    double* o1 = (double*)workspace;
    double* o2 = o1 + m1 * T;
    const bool logm = metric == GABO_METRIC_LOG_EUCLIDEAN;
#define GABO_CASE(DL_)                                                                                                              \
    case DL_:                                                                                                                       \
        if (logm) gabo::launch_nested_prepare<DL_, true>(x1, x2, w, o1, o2, m1, m2, n2, D, same, status, st);                        \
        else gabo::launch_nested_prepare<DL_, false>(x1, x2, w, o1, o2, m1, m2, n2, D, same, status, st);                            \
        break;
    switch (dl) { GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) }
#undef GABO_CASE
    if (hipGetLastError() != hipSuccess) return GABO_ERR_LAUNCH;
    if (logm) {
        const double* f2 = same ? o1 : o2;
        return gabo_frobenius_pairwise(o1, f2, out, batch, n1, n2, dl, n1 * T, n2 * T, beta, flags & GABO_OUT_MASK, stream);
    }
    return gabo::launch_spd_ai_prepared(dl, out, batch, n1, n2, false, false, beta, flags, o1, st);
}
