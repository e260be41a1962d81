// Fused exact-GP posterior + analytic acquisition (value and gradient with respect to the cross-covariance strip), one block
// per candidate.  This is the step that consumes the kernel strip K(X*, X_train) in the acquisition maximiser
// (manifold_optimize.py:182-184 -> acquisition(X) -> GP posterior -> ExpectedImprovement; [3P] botorch/gpytorch semantics
// restated in gabotorch_amd/models.py and SURVEY App. B):
//   ks    = outputscale * kstar[r, :]
//   mean  = m + ks . alpha                          alpha = (K + noise I)^-1 (y - m)
//   v     = L^-1 ks                                 L = chol(K + noise I)
//   var   = outputscale * kxx - v . v
//   EI    = sigma (phi(u) + u Phi(u)),  sigma = sqrt(max(var, 1e-9)),  u = +-(mean - best_f) / sigma
// and dAcq/dkstar[r, j] = outputscale (s Phi(u) alpha_j - [var > 1e-9] phi(u) / sigma * (L^-T v)_j).
// The reference leaves all of this to autograd over a few dozen tiny torch kernels per evaluation; here it is one launch.
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

static __device__ __forceinline__ double block_sum(double v, double* red) {
    // blockDim.x is a multiple of 64 (<= 256)
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += red[k];
    return s;
}

__global__ __launch_bounds__(256) void gp_acquisition_kernel(const double* __restrict__ kstar, const double* __restrict__ alpha,
                                                             const double* __restrict__ linv, const double* __restrict__ linv_t,
                                                             double* __restrict__ value, double* __restrict__ grad_k, int64_t n,
                                                             double mean0, double os, double kxx, double best_f, int kind,
                                                             int maximize, double out_sign) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* ks = lds;         // n
    double* v = ks + n;       // n
    double* red = v + n;      // 4
    const int64_t r = blockIdx.x;
    const double* kr = kstar + r * n;
    double part = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
        double k = os * kr[j];
        ks[j] = k;
        part = __builtin_fma(k, alpha[j], part);
    }
    const double mean = mean0 + block_sum(part, red);      // (the barrier inside also publishes ks)
    const double s = maximize ? 1.0 : -1.0;
    if (kind == GABO_ACQ_POSTERIOR_MEAN) {
        if (threadIdx.x == 0) value[r] = out_sign * s * mean;
        if (grad_k)
            for (int64_t j = threadIdx.x; j < n; j += blockDim.x) grad_k[r * n + j] = out_sign * s * os * alpha[j];
        return;
    }
    // v = L^-1 ks : thread i owns row i, walks the columns j <= i of L^-1 through its transpose (coalesced across i)
    part = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        double a = 0.0;
        for (int64_t j = 0; j <= i; ++j) a = __builtin_fma(linv_t[j * n + i], ks[j], a);
        v[i] = a;
        part = __builtin_fma(a, a, part);
    }
    const double var = os * kxx - block_sum(part, red);
    const bool clamped = !(var > 1e-9);
    const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
    const double u = s * (mean - best_f) / sigma;
    const double pdf = exp(-0.5 * u * u) * 0.3989422804014327;      // 1/sqrt(2 pi)
    const double cdf = 0.5 * (1.0 + erf(u * 0.7071067811865476));
    if (threadIdx.x == 0) value[r] = out_sign * sigma * (pdf + u * cdf);
    if (!grad_k) return;
    const double g_mean = s * cdf;                         // dEI/dmean
    const double g_var = clamped ? 0.0 : 0.5 * pdf / sigma;   // dEI/dvar = phi(u) / (2 sigma)
    // w = L^-T v : thread j owns column j of L^-1 (rows i >= j), coalesced across j
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
        double w = 0.0;
        for (int64_t i = j; i < n; ++i) w = __builtin_fma(linv[i * n + j], v[i], w);
        grad_k[r * n + j] = out_sign * os * (g_mean * alpha[j] - 2.0 * g_var * w);
    }
}

}  // namespace gabo

extern "C" int gabo_gp_acquisition(const double* kstar, const double* alpha, const double* linv, const double* linv_t, double* value,
                                   double* grad_kstar, int64_t r, int64_t n, double mean, double outputscale, double kxx,
                                   double best_f, int kind, int maximize, double out_sign, gabo_stream_t stream) {
    if (r < 0 || n < 1 || n > 4096) return GABO_ERR_ARG;
    if (kind != GABO_ACQ_EXPECTED_IMPROVEMENT && kind != GABO_ACQ_POSTERIOR_MEAN) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!kstar || !alpha || !value || r > 0x7fffffffLL) return GABO_ERR_ARG;
    if (kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!linv || !linv_t)) return GABO_ERR_ARG;
    const int threads = n <= 64 ? 64 : (n <= 128 ? 128 : 256);
    size_t lds = (size_t)(2 * n + 4) * sizeof(double);
    hipLaunchKernelGGL(gabo::gp_acquisition_kernel, dim3((unsigned)r), dim3(threads), lds, (hipStream_t)stream, kstar, alpha, linv,
                       linv_t, value, grad_kstar, n, mean, outputscale, kxx, best_f, kind, maximize, out_sign);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
