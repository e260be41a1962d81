// Per-matrix preparation shared by the forward and backward SPD kernels: Mandel vector -> Cholesky factor.
//   x1 side: W = L^-1, packed lower, row contiguous  [b][n1][T]   (wave-uniform operand, read through the scalar cache)
//   x2 side: G = chol(B), entry-major "SoA"           [b][T][n2]   (lane j reads G[e][j]: fully coalesced)
// A non-positive pivot (input not SPD; the reference raises from torch.cholesky, spd_utils_torch.py:87) sets
// status[0] = GABO_ERR_NOT_SPD and status[1] = index of the first offender (x1 set first, then x2).
// `lenient2` (the forward Gram): the reference factors x1 only - a matrix of x2 that is not positive definite but free of NaN goes through its
// symeig and log and gives a NaN column (spd_utils_torch.py:109-120), no exception; NaN entries in x2 make its eigen-solver raise.  With the flag
// such a matrix of x2 is not reported: its factor carries the NaN of the failed pivot into every entry of its column of the result.
#pragma once
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

// Mandel vector -> packed lower Cholesky factor a (in registers); returns true when a pivot is not positive
template <int D>
__device__ __forceinline__ bool mandel_cholesky(const double* __restrict__ v, double (&a)[tri_size(D)]) {
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            double e = v[mandel_pos(D, r, c)];
            // (spd_utils_torch.py:186-187 divides by 2**0.5; the product with 1/sqrt2 differs from the quotient by at most one ulp of an entry that
            // only feeds the factorisation, and an fp64 division is ~35 instructions: T of them per matrix were the larger part of this function)
            a[tri(r, c)] = (r == c) ? e : e * kInvSqrt2;
        });
    });
    bool bad = false;
    static_for<D>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double piv = a[tri(c, c)];
        static_for<c>([&](auto kk) { constexpr int k = decltype(kk)::value; piv = __builtin_fma(-a[tri(c, k)], a[tri(c, k)], piv); });
        if (!(piv > 0.0)) bad = true;
        // 1 / sqrt(piv) by seed + cubic correction, L_cc = piv / sqrt(piv): 8 instructions for the IEEE sqrt and division's ~50 (a non-positive
        // pivot - flagged above - still yields NaN / inf here, as the square root did)
        double inv = rsqrt_nz(piv);
        double lcc = piv * inv;
        a[tri(c, c)] = lcc;
        static_for<D - c - 1>([&](auto rr) {
            constexpr int r = c + 1 + decltype(rr)::value;
            double s = a[tri(r, c)];
            static_for<c>([&](auto kk) { constexpr int k = decltype(kk)::value; s = __builtin_fma(-a[tri(r, k)], a[tri(c, k)], s); });
            a[tri(r, c)] = s * inv;
        });
    });
    return bad;
}

// W = L^-1 (lower), column by column: W[c][c] = 1/L[c][c]; W[r][c] = -(sum_{k=c}^{r-1} L[r][k] W[k][c]) / L[r][r]
template <int D>
__device__ __forceinline__ void lower_inverse(const double (&a)[tri_size(D)], double (&w)[tri_size(D)]) {
    static_for<D>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        w[tri(c, c)] = rcp(a[tri(c, c)]);
    });
    static_for<D>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        static_for<D - c - 1>([&](auto rr) {
            constexpr int r = c + 1 + decltype(rr)::value;
            double s = 0.0;
            static_for<r - c>([&](auto kk) {
                constexpr int k = c + decltype(kk)::value;
                s = __builtin_fma(a[tri(r, k)], w[tri(k, c)], s);
            });
            w[tri(r, c)] = -s * w[tri(r, r)];
        });
    });
}

// One launch factors BOTH point sets: blocks [0, blocks1) handle x1 (W = chol(x1)^-1, row contiguous), the rest x2 (G = chol(x2),
// entry-major).  (Two launches of ~6 us each were 0.5 % of a d = 10, N = 4096 Gram build.)
template <int D>
__global__ __launch_bounds__(64) void spd_prep_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                      double* __restrict__ W, double* __restrict__ G, int64_t b1, int64_t b2,
                                                      int64_t n1, int64_t n2, int64_t s1, int64_t s2, unsigned blocks1,
                                                      int* __restrict__ status, int lenient2) {
    constexpr int T = tri_size(D);
    const bool second = blockIdx.x >= blocks1;
    const int64_t g = (int64_t)(second ? blockIdx.x - blocks1 : blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t n = second ? n2 : n1;
    if (g >= (second ? b2 : b1) * n) return;
    const int64_t b = g / n, i = g - b * n;
    const double* v = (second ? x2 + b * s2 : x1 + b * s1) + i * T;
    double a[T];
    bool bad = mandel_cholesky<D>(v, a);
    if (bad && second && lenient2) {
        bool has_nan = false;
        static_for<T>([&](auto ee) { const double e = v[decltype(ee)::value]; has_nan |= e != e; });
        bad = has_nan;
        // every entry of the factor NaN (a failed LAST pivot alone would leave the other columns of the factor finite)
        if (!has_nan) static_for<T>([&](auto ee) { a[decltype(ee)::value] = __builtin_nan(""); });
    }
    if (bad) {
        if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (second ? (int)(b1 * n1) : 0) + (int)g;
    }
    if (second) {
        // G, entry-major: G[(b*T + e) * n + i]
        static_for<T>([&](auto ee) { G[(b * T + decltype(ee)::value) * n + i] = a[decltype(ee)::value]; });
    } else {
        double w[T];
        lower_inverse<D>(a, w);
        double* o = W + g * T;
        static_for<T>([&](auto ee) { o[decltype(ee)::value] = w[decltype(ee)::value]; });
    }
}

template <int D>
static void launch_spd_prep(const double* x1, const double* x2, double* W, double* G, int64_t b1, int64_t b2, int64_t n1,
                            int64_t n2, int64_t s1, int64_t s2, int* status, hipStream_t st, int lenient2 = 0) {
    const unsigned blocks1 = (unsigned)((b1 * n1 + 63) / 64), blocks2 = (unsigned)((b2 * n2 + 63) / 64);
    if (blocks1 + blocks2 == 0) return;
    hipLaunchKernelGGL((spd_prep_kernel<D>), dim3(blocks1 + blocks2), dim3(64), 0, st, x1, x2, W, G, b1, b2, n1, n2, s1, s2, blocks1,
                       status, lenient2);
}

}  // namespace gabo
