// Prediction cache of the exact GP in one launch: L = chol(outputscale K + noise I), L^-1, L^-T and alpha = (outputscale K + noise I)^-1 (y - mean).
// This is what [3P] gpytorch computes when a fitted ExactGP is first asked for a posterior (the `prediction_strategy` caches behind
// manifold_optimize.py:182-184), and what every acquisition sweep of this package starts from (fused_acquisition.py: gabo_gp_acquisition and the
// single-launch solves read alpha, L^-1 and L^-T).  Through torch it was a Cholesky (with its device -> host info read-back), a cholesky_solve, a
// triangular solve against the identity and a transposed copy: ~0.5 ms of a 4.4-ms config-4 sweep (tools/ic_phases.py) for a 50 x 50 matrix.
// Latency design like gp_mll.hip: ONE workgroup, the matrix and its inverse factor in LDS (n <= GABO_GP_FACTOR_MAX_N = 96: 2 n^2 doubles),
// left-looking Cholesky (row r of column c is one thread's dot product, four accumulators), column-per-thread forward substitution for L^-1.
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

__global__ __launch_bounds__(256) void gp_factor_kernel(const double* __restrict__ kb, const double* __restrict__ y, int n, double outputscale,
                                                        double noise, double mean, double* __restrict__ linv, double* __restrict__ linv_t,
                                                        double* __restrict__ alpha, int* __restrict__ status) {
    extern __shared__ double sm[];
    double* A = sm;              // n x n: the matrix, then L (strict upper zeroed)
    double* W = A + n * n;       // n x n: L^-1
    double* v = W + n * n;       // n: L^-1 r
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        A[e] = __builtin_fma(outputscale, kb[e], i == j ? noise : 0.0);
    }
    __syncthreads();
    bool ok = true;
    for (int c = 0; c < n; ++c) {
        // every thread recomputes the pivot (wave-uniform data, no broadcast needed)
        double p0 = A[c * n + c], p1 = 0.0, p2 = 0.0, p3 = 0.0;
        int k = 0;
        for (; k + 4 <= c; k += 4) {
            const double a0 = A[c * n + k], a1 = A[c * n + k + 1], a2 = A[c * n + k + 2], a3 = A[c * n + k + 3];
            p0 = __builtin_fma(-a0, a0, p0);
            p1 = __builtin_fma(-a1, a1, p1);
            p2 = __builtin_fma(-a2, a2, p2);
            p3 = __builtin_fma(-a3, a3, p3);
        }
        for (; k < c; ++k) p0 = __builtin_fma(-A[c * n + k], A[c * n + k], p0);
        const double piv = (p0 + p1) + (p2 + p3);
        if (!(piv > 0.0)) ok = false;
        const double inv = rsqrt_nz(ok ? piv : 1.0);
        __syncthreads();                 // (everybody has read row c before its diagonal entry is overwritten)
        for (int r = c + tid; r < n; r += nt) {
            if (r == c) {
                A[c * n + c] = piv * inv;
            } else {
                double s0 = A[r * n + c], s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int q = 0;
                for (; q + 4 <= c; q += 4) {
                    s0 = __builtin_fma(-A[r * n + q], A[c * n + q], s0);
                    s1 = __builtin_fma(-A[r * n + q + 1], A[c * n + q + 1], s1);
                    s2 = __builtin_fma(-A[r * n + q + 2], A[c * n + q + 2], s2);
                    s3 = __builtin_fma(-A[r * n + q + 3], A[c * n + q + 3], s3);
                }
                for (; q < c; ++q) s0 = __builtin_fma(-A[r * n + q], A[c * n + q], s0);
                A[r * n + c] = ((s0 + s1) + (s2 + s3)) * inv;
            }
        }
        for (int r = tid; r < c; r += nt) A[r * n + c] = 0.0;
        __syncthreads();
    }
    if (!ok) {
        if (tid == 0) { status[0] = GABO_ERR_NOT_SPD; status[1] = 0; }
        return;       // (wave-uniform: `ok` is computed identically by every thread)
    }
    // W = L^-1: thread c owns column c (forward substitution); rows above the diagonal are exact zeros
    for (int c = tid; c < n; c += nt) {
        for (int r = 0; r < c; ++r) W[r * n + c] = 0.0;
        W[c * n + c] = rcp(A[c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = c;
            for (; k + 4 <= r; k += 4) {
                s0 = __builtin_fma(A[r * n + k], W[k * n + c], s0);
                s1 = __builtin_fma(A[r * n + k + 1], W[(k + 1) * n + c], s1);
                s2 = __builtin_fma(A[r * n + k + 2], W[(k + 2) * n + c], s2);
                s3 = __builtin_fma(A[r * n + k + 3], W[(k + 3) * n + c], s3);
            }
            for (; k < r; ++k) s0 = __builtin_fma(A[r * n + k], W[k * n + c], s0);
            W[r * n + c] = -((s0 + s1) + (s2 + s3)) * rcp(A[r * n + r]);
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {          // v = L^-1 (y - mean)
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s = __builtin_fma(W[i * n + k], y[k] - mean, s);
        v[i] = s;
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) {          // alpha = L^-T v
        double s = 0.0;
        for (int i = j; i < n; ++i) s = __builtin_fma(W[i * n + j], v[i], s);
        alpha[j] = s;
    }
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        const double w = W[e];
        linv[e] = w;
        linv_t[j * n + i] = w;
    }
}

}  // namespace gabo

extern "C" int gabo_gp_factor(const double* k, const double* y, int64_t n, double outputscale, double noise, double mean, double* linv,
                              double* linv_t, double* alpha, int* status, gabo_stream_t stream) {
    if (n < 0 || !k || !y || !linv || !linv_t || !alpha || !status) return GABO_ERR_ARG;
    if (n > GABO_GP_FACTOR_MAX_N) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    const size_t lds = (size_t)(2 * n * n + n) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        // (more than 64 KB of dynamic LDS needs the attribute; set once, the maximum the entry point accepts)
        if (hipFuncSetAttribute((const void*)gabo::gp_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((2 * GABO_GP_FACTOR_MAX_N * GABO_GP_FACTOR_MAX_N + GABO_GP_FACTOR_MAX_N) * sizeof(double))) != hipSuccess)
            return GABO_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(gabo::gp_factor_kernel, dim3(1), dim3(n <= 64 ? 64 : (n <= 128 ? 128 : 256)), lds, (hipStream_t)stream, k, y, (int)n,
                       outputscale, noise, mean, linv, linv_t, alpha, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
