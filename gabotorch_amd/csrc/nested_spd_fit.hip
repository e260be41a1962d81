// One evaluation of HD-GaBO's surrogate-fit objective with its gradient, as ONE host call:
//   marginal log likelihood of  ScaleKernel(NestedSpdLogEuclideanGaussianKernel)  at a projection matrix W in G(D, d) and scalar
//   hyper-parameters, with d ll / d(theta, outputscale, noise, mean) and d ll / d W
//   (fit_gpytorch_manifold's closure, manifold_optimization/manifold_gp_fit.py:54-222; the kernel: kernel_utils/kernels_nested_spd.py:139-250 =
//    projection_from_spd_to_nested_spd (nested_spd_utils.py:20-48) -> logm_torch per point (spd_utils_torch.py:13-30) -> Gaussian of the
//    Frobenius distances (:124-156); the likelihood: [3P] gpytorch ExactMarginalLogLikelihood; the reference differentiates all of it by autograd).
// The pieces exist as separate entry points (gabo_spd_project, gabo_spd_logm_mandel, gabo_frobenius_pairwise, gabo_gp_mll_gram /
// gabo_gp_mll_large and their backward launches); round 3's Python chain issued them through ~16 ctypes / torch calls per gradient
// (~0.2 ms of host time for ~60 us of kernels, 165 evaluations per fit).  Here the same launches are issued back to back from C++ on the
// caller's stream, with two small kernels of this file for what torch did in between (the adjoint of the Gram matrix and d/d theta;
// the projection's adjoint dW = 2 sum_n X_n W G_n), one pinned copy in (W) and one out (7 + D d doubles).
#include <hip/hip_runtime.h>

#include <cstring>

#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

// gs = (outputscale / 2) (W + W^T)  (d ll / d kb, both arguments of the symmetric Gram matrix at once), and
// out[6] = d ll / d theta = sum_ij (outputscale / 2) W_ij kb_ij log(kb_ij) / theta   (kb = exp(-theta E): d kb / d theta = kb log(kb) / theta).
// Grid-stride over the n^2 entries; per-block partial sums in `partial`, added in block order by the last block to finish.
__global__ __launch_bounds__(256) void fit_gram_adjoint_kernel(const double* __restrict__ kb, const double* __restrict__ wm,
                                                               double* __restrict__ gs, double* __restrict__ out, double* __restrict__ partial,
                                                               int* __restrict__ counter, int64_t n, double half_os, double theta) {
    __shared__ double red[4];
    __shared__ int last;
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / n, j = e - i * n;
        const double w = wm[e], k = kb[e];
        gs[e] = half_os * (w + wm[j * n + i]);
        if (k > 0.0) acc = __builtin_fma(half_os * w, k * log(k), acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __threadfence();
    double s = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) s += partial[b];
    out[6] = s / theta;
}

// gw = 2 sum_n X_n W G_n,  G_n = the matrix of the Mandel vector gz_n (d x d, off-diagonal entries / sqrt 2): the adjoint of
// Y_n = W^T X_n W.  Block b takes the points n = b, b + gridDim, ...; partial sums per block, added in block order by the last one.
__global__ __launch_bounds__(256) void fit_projection_adjoint_kernel(const double* __restrict__ xm, const double* __restrict__ w,
                                                                     const double* __restrict__ gz, double* __restrict__ partial,
                                                                     double* __restrict__ gw, int* __restrict__ counter, int64_t n, int D, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Wl = lds;                 // D x d
    double* G = Wl + D * d;           // d x d
    double* T = G + d * d;            // D x d : W G_n
    __shared__ int last;
    const int dv = d * (d + 1) / 2, Dd = D * d;
    for (int e = threadIdx.x; e < Dd; e += blockDim.x) Wl[e] = w[e];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};                 // D d <= 32 * 31 < 4 * 256 outputs
    for (int64_t q = blockIdx.x; q < n; q += gridDim.x) {
        __syncthreads();
        for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
            const int r = e / d, c = e - r * d;
            const double v = gz[q * dv + mandel_pos(d, r > c ? r : c, r > c ? c : r)];
            G[e] = r == c ? v : v / kSqrt2;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < Dd; e += blockDim.x) {
            const int r = e / d, c = e - r * d;
            double s = 0.0;
            for (int k = 0; k < d; ++k) s = __builtin_fma(Wl[r * d + k], G[k * d + c], s);
            T[e] = s;
        }
        __syncthreads();
        const double* X = xm + q * D * D;
        for (int t = 0; t < 4; ++t) {
            const int e = threadIdx.x + t * blockDim.x;
            if (e < Dd) {
                const int r = e / d, c = e - r * d;
                double s = acc[t];
                for (int k = 0; k < D; ++k) s = __builtin_fma(X[r * D + k], T[k * d + c], s);
                acc[t] = s;
            }
        }
    }
    for (int t = 0; t < 4; ++t) {
        const int e = threadIdx.x + t * blockDim.x;
        if (e < Dd) partial[(size_t)blockIdx.x * Dd + e] = acc[t];
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    for (int e = threadIdx.x; e < Dd; e += blockDim.x) {
        double s = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) s += partial[(size_t)b * Dd + e];
        gw[e] = 2.0 * s;
    }
}

// (for the nested-sphere chain in nested_sphere_chain.hip: the same adjoint of the Gram matrix)
int fit_gram_adjoint_launch(const double* kb, const double* wm, double* gs, double* out, double* partial, int* counter, int64_t n, double half_os,
                            double theta, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(fit_gram_adjoint_kernel, dim3(blocks), dim3(256), 0, s, kb, wm, gs, out, partial, counter, n, half_os, theta);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

namespace {

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct FitLayout {
    size_t w, z, feat, kb, wm, gs, gfeat, gz, out, partial, counters, mll, total;
    unsigned blocks_gram, blocks_proj;
    FitLayout(int64_t n, int D, int d) {
        const size_t dv = (size_t)d * (d + 1) / 2, nn = (size_t)n * n;
        blocks_gram = (unsigned)((nn + 255) / 256 > 256 ? 256 : (nn + 255) / 256);
        blocks_proj = (unsigned)(n > 64 ? 64 : (n < 1 ? 1 : n));
        size_t o = 0;
        auto take = [&o](size_t doubles) { const size_t at = o; o += align256(doubles * sizeof(double)); return at; };
        w = take((size_t)D * d);
        z = take((size_t)n * dv);
        feat = take((size_t)n * dv);
        kb = take(nn);
        wm = take(nn);
        gs = take(nn);
        gfeat = take((size_t)n * dv);
        gz = take((size_t)n * dv);
        out = take(7 + (size_t)D * d);                     // [ll, 0, d os, d noise, d mean, flag, d theta, gW (D x d)]
        const size_t pg = blocks_gram, pp = (size_t)blocks_proj * D * d;
        partial = take(pg > pp ? pg : pp);
        counters = take(2);
        mll = o;
        o += align256(n > GABO_GP_MLL_MAX_N ? gabo_gp_mll_large_workspace_bytes(n) : 0);
        total = o;
    }
};

}  // namespace
}  // namespace gabo

extern "C" {

size_t gabo_nested_spd_fit_workspace_bytes(int64_t n, int D, int d) {
    if (n < 1 || D < 2 || d < 1 || d >= D) return 0;
    return gabo::FitLayout(n, D, d).total;
}

int gabo_nested_spd_fit_evaluate(const double* x_mandel, const double* x_matrices, const double* y, const double* w_host, int64_t n, int D,
                                 int d, double theta, double outputscale, double noise, double mean, int want_grad, double* out_host,
                                 void* workspace, size_t workspace_bytes, double* pinned, size_t pinned_doubles, gabo_stream_t stream) {
    if (D < 2 || D > GABO_SPD_MAX_DIM || d < 1 || d >= D) return GABO_ERR_DIM;
    if (n < 1 || n > GABO_GP_MLL_LARGE_MAX_N) return GABO_ERR_DIM;
    const size_t Dd = (size_t)D * d;
    if (!x_mandel || !x_matrices || !y || !w_host || !out_host || !workspace || !pinned || !(theta > 0.0)) return GABO_ERR_ARG;
    const gabo::FitLayout lay(n, D, d);
    if (workspace_bytes < lay.total || pinned_doubles < Dd + 7 + Dd) return GABO_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    char* base = static_cast<char*>(workspace);
    auto at = [base](size_t off) { return reinterpret_cast<double*>(base + off); };
    double* d_w = at(lay.w);
    double* d_out = at(lay.out);
    int* counters = reinterpret_cast<int*>(base + lay.counters);
    if (want_grad && hipMemsetAsync(counters, 0, 2 * sizeof(int), s) != hipSuccess) return GABO_ERR_LAUNCH;     // the two tickets
    std::memcpy(pinned, w_host, sizeof(double) * Dd);
    if (hipMemcpyAsync(d_w, pinned, sizeof(double) * Dd, hipMemcpyHostToDevice, s) != hipSuccess) return GABO_ERR_LAUNCH;
    int rc = gabo_spd_project(x_mandel, d_w, at(lay.z), n, D, d, stream);
    if (rc == GABO_OK) rc = gabo_spd_logm_mandel(at(lay.z), at(lay.feat), n, d, stream);
    if (rc == GABO_OK) rc = gabo_frobenius_pairwise(at(lay.feat), at(lay.feat), at(lay.kb), 1, n, n, d, 0, 0, theta, GABO_OUT_GAUSSIAN, stream);
    if (rc != GABO_OK) return rc;
    double* wm = want_grad ? at(lay.wm) : nullptr;
    if (n <= GABO_GP_MLL_MAX_N)
        rc = gabo_gp_mll_gram(at(lay.kb), y, n, outputscale, noise, mean, d_out, wm, stream);
    else
        rc = gabo_gp_mll_large(at(lay.kb), y, n, 0.0, outputscale, noise, mean, 1, d_out, wm, base + lay.mll, gabo_gp_mll_large_workspace_bytes(n), stream);
    if (rc != GABO_OK) return rc;
    size_t out_doubles = 6;
    if (want_grad) {
        hipLaunchKernelGGL(gabo::fit_gram_adjoint_kernel, dim3(lay.blocks_gram), dim3(256), 0, s, at(lay.kb), wm, at(lay.gs), d_out, at(lay.partial),
                           counters, n, 0.5 * outputscale, theta);
        if (hipGetLastError() != hipSuccess) return GABO_ERR_LAUNCH;
        // both arguments of the symmetric Gram matrix at once: gs is symmetrised, so the x1-gradient with it is the whole gradient
        // (up to the 1e-15 the reference adds to the difference, which enters the two arguments with opposite signs)
        rc = gabo_frobenius_backward(at(lay.feat), at(lay.feat), at(lay.gs), at(lay.gfeat), 1, n, n, d, 0, 0, n * n, n, 1, theta, GABO_OUT_GAUSSIAN,
                                     1.0, stream);
        if (rc == GABO_OK) rc = gabo_spd_logm_mandel_backward(at(lay.z), at(lay.gfeat), at(lay.gz), n, d, stream);
        if (rc != GABO_OK) return rc;
        const size_t lds = (Dd + (size_t)d * d + Dd) * sizeof(double);
        hipLaunchKernelGGL(gabo::fit_projection_adjoint_kernel, dim3(lay.blocks_proj), dim3(256), lds, s, x_matrices, d_w, at(lay.gz), at(lay.partial),
                           d_out + 7, counters + 1, n, D, d);
        if (hipGetLastError() != hipSuccess) return GABO_ERR_LAUNCH;
        out_doubles = 7 + Dd;
    }
    double* h_out = pinned + Dd;
    if (hipMemcpyAsync(h_out, d_out, sizeof(double) * out_doubles, hipMemcpyDeviceToHost, s) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return GABO_ERR_LAUNCH;
    std::memcpy(out_host, h_out, sizeof(double) * out_doubles);
    if (!want_grad) {
        out_host[6] = 0.0;
        std::memset(out_host + 7, 0, sizeof(double) * Dd);
    }
    return GABO_OK;
}

}  // extern "C"
