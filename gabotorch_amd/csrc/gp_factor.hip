// Prediction cache of the exact GP in one launch: L = chol(outputscale K + noise I), L^-1, L^-T and alpha = (outputscale K + noise I)^-1 (y - mean).
// This is what [3P] gpytorch computes when a fitted ExactGP is first asked for a posterior (the `prediction_strategy` caches behind
// manifold_optimize.py:182-184), and what every acquisition sweep of this package starts from (fused_acquisition.py: gabo_gp_acquisition and the
// single-launch solves read alpha, L^-1 and L^-T).  Through torch it was a Cholesky (with its device -> host info read-back), a cholesky_solve, a
// triangular solve against the identity and a transposed copy: ~0.5 ms of a 4.4-ms config-4 sweep (tools/ic_phases.py) for a 50 x 50 matrix.
// Latency design: ONE workgroup of 256 threads as a 16 x 16 grid over the matrix in LDS (n <= GABO_GP_FACTOR_MAX_N = 96: 2 n^2 + 3 n doubles).
// RIGHT-LOOKING elimination, L^-1 carried along: step k scales column k of the trailing matrix into l_k and row k of W = [L^-1 | L^-1 (y - mean)],
// then every thread applies the rank-one updates A[r, c] -= l_r l_c (k < c <= r) and W[r, :] -= l_r W[k, :] (r > k) to its own entries - all
// independent, two barriers per step.  (Rounds 2-5: left-looking, row r of column c one thread's dot product, then column-per-thread forward
// substitution - every step a dependent chain of LDS reads: 104 us at n = 50 under rocprofv3, the longest launch of the sweep's set-up.)
#include <atomic>

#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

__global__ __launch_bounds__(256) void gp_factor_kernel(const double* __restrict__ kb, const double* __restrict__ y, int n, double outputscale,
                                                        double noise, double mean, double* __restrict__ linv, double* __restrict__ linv_t,
                                                        double* __restrict__ alpha, double* __restrict__ kinv, int* __restrict__ status) {
    extern __shared__ double sm[];
    const int nw = n + 1;
    double* A = sm;              // n x n: the matrix; its trailing block is updated in place (lower triangle read)
    double* W = A + n * n;       // n x (n + 1): rows of L^-1, column n = L^-1 (y - mean)
    double* lk = W + n * nw;     // n: column k of L below the diagonal
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tx = tid & 15, ty = tid >> 4;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        A[e] = __builtin_fma(outputscale, kb[e], i == j ? noise : 0.0);
        W[i * nw + j] = i == j ? 1.0 : 0.0;
    }
    for (int i = tid; i < n; i += nt) W[i * nw + n] = y[i] - mean;
    __syncthreads();
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        const double piv = A[k * n + k];              // (the same value in every thread)
        if (!(piv > 0.0)) ok = false;
        const double inv = rsqrt_nz(ok ? piv : 1.0);   // 1 / L[k, k]
        for (int r = k + 1 + tid; r < n; r += nt) lk[r] = A[r * n + k] * inv;
        for (int c = tid; c <= k; c += nt) W[k * nw + c] *= inv;
        if (tid == nt - 1) W[k * nw + n] *= inv;
        __syncthreads();
        for (int r = k + 1 + ty; r < n; r += 16) {
            const double lr = lk[r];
            for (int c = k + 1 + tx; c <= r; c += 16) A[r * n + c] = __builtin_fma(-lr, lk[c], A[r * n + c]);
            for (int c = tx; c <= k; c += 16) W[r * nw + c] = __builtin_fma(-lr, W[k * nw + c], W[r * nw + c]);
            if (tx == 15) W[r * nw + n] = __builtin_fma(-lr, W[k * nw + n], W[r * nw + n]);
        }
        __syncthreads();
    }
    if (!ok) {
        if (tid == 0) { status[0] = GABO_ERR_NOT_SPD; status[1] = 0; }
        return;       // (uniform: `ok` is computed identically by every thread)
    }
    for (int j = tid; j < n; j += nt) {          // alpha = L^-T v
        double s = 0.0;
        for (int i = j; i < n; ++i) s = __builtin_fma(W[i * nw + j], W[i * nw + n], s);
        alpha[j] = s;
    }
    for (int e = tid; e < n * n; e += nt) {      // (rows above the diagonal of W were never touched: exact zeros)
        const int i = e / n, j = e - i * n;
        const double w = W[i * nw + j];
        linv[e] = w;
        linv_t[j * n + i] = w;
    }    if (kinv != nullptr) {
        // A = (outputscale K + noise I)^-1 = L^-T L^-1: A[i][j] = sum_{k >= i} W[k][i] W[k][j] for i >= j, mirrored (exactly symmetric).  With it the
        // posterior variance and its gradient need ONE n-term product per training point instead of the two triangular ones (spd_acq_body.hpp).
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            if (i < j) continue;
            double s0 = 0.0, s1 = 0.0;
            int k = i;
            for (; k + 2 <= n; k += 2) {
                s0 = __builtin_fma(W[k * nw + i], W[k * nw + j], s0);
                s1 = __builtin_fma(W[(k + 1) * nw + i], W[(k + 1) * nw + j], s1);
            }
            if (k < n) s0 = __builtin_fma(W[k * nw + i], W[k * nw + j], s0);
            const double a = s0 + s1;
            kinv[i * n + j] = a;
            kinv[j * n + i] = a;
        }
    }
}


}  // namespace gabo

extern "C" int gabo_gp_factor(const double* k, const double* y, int64_t n, double outputscale, double noise, double mean, double* linv,
                              double* linv_t, double* alpha, double* kinv, int* status, gabo_stream_t stream) {
    if (n < 0 || !k || !y || !linv || !linv_t || !alpha || !status) return GABO_ERR_ARG;
    if (n > GABO_GP_FACTOR_MAX_N) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    const size_t lds = (size_t)(2 * n * n + 3 * n) * sizeof(double);
    // (more than 64 KB of dynamic LDS needs the attribute: set once per DEVICE, the maximum the entry point accepts)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return GABO_ERR_LAUNCH;
    if (!(attr_set.load(std::memory_order_acquire) >> dev & 1)) {
        if (hipFuncSetAttribute((const void*)gabo::gp_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((2 * GABO_GP_FACTOR_MAX_N * GABO_GP_FACTOR_MAX_N + 3 * GABO_GP_FACTOR_MAX_N) * sizeof(double))) != hipSuccess)
            return GABO_ERR_LAUNCH;
        attr_set.fetch_or((uint64_t)1 << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(gabo::gp_factor_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, k, y, (int)n, outputscale, noise, mean, linv, linv_t,
                       alpha, kinv, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
