// Exact-GP marginal log likelihood and its analytic gradient with respect to the kernel hyper-parameters, one launch per
// evaluation.  This is the inner step of the surrogate fit that runs every BO iteration (examples/.../gabo_spd.py:194
// `fit_gpytorch_model(mll)` -> [3P] gpytorch ExactMarginalLogLikelihood under scipy L-BFGS-B).  For every plain kernel of the path
// the Gram matrix has the form exp(-theta * E) with E = d^2 (Gaussian) or d (Laplace) fixed during the fit, so the pairwise
// distances are evaluated ONCE (gabo_spd_ai_pairwise / gabo_frobenius_pairwise / gabo_sphere_pairwise with GABO_OUT_DISTANCE) and
// each L-BFGS evaluation is the small dense algebra below:
//   Ky = outputscale * exp(-theta E) + noise I,  r = y - mean,  alpha = Ky^-1 r,  W = alpha alpha^T - Ky^-1
//   ll         = -r.alpha/2 - log det(Ky)/2 - n log(2 pi)/2
//   dll/dtheta = tr(W dKy/dtheta)/2,  dKy/dtheta = -outputscale * E o exp(-theta E)
//   dll/dos    = tr(W exp(-theta E))/2,  dll/dnoise = tr(W)/2,  dll/dmean = sum_i alpha_i
// The reference leaves this to autograd through the per-pair Python loop of the kernel on every evaluation.
//
// The problem is one small matrix, so the design target is latency, not throughput: one workgroup, the bordered matrix
// [[Ky, r], [r^T, 0]] (packed lower triangle) distributed over the threads' REGISTERS, and the symmetric sweep operator
// (Gauss-Jordan on the pivots 0..n-1: a_kk <- -1/p, a_ik <- a_ik/p, a_ij <- a_ij - a_ik a_jk/p) applied in place.  After the n
// sweeps the matrix block holds -Ky^-1, the border holds alpha and the corner -r.alpha; the pivots are the Schur complements, so
// log det(Ky) = sum log p_k and Ky is positive definite iff every p_k > 0.  One sweep is one barrier: only the pivot column
// travels through LDS (double-buffered: the owners of the next pivot's row/column publish it while they update it).
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

static __device__ __forceinline__ int ltri_i(int r) { return r * (r + 1) / 2; }

template <int THREADS>
static __device__ __forceinline__ double mll_block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < THREADS / 64; ++k) s += red[k];
    return s;
}

// MAXE: packed entries of the (n+1) x (n+1) bordered matrix owned by one thread (entry idx = e * THREADS + t)
template <int THREADS, int MAXE>
__global__ __launch_bounds__(THREADS) void gp_mll_kernel(const double* __restrict__ e, const double* __restrict__ y, int n,
                                                         double theta, double os, double noise, double mean,
                                                         double* __restrict__ out, int gram, double* __restrict__ w_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = n + 1;
    double* col0 = lds;           // m + 1: the pivot column of the current sweep (entry k is the pivot itself; entry m is padding)
    double* col1 = col0 + m + 1;  // m + 1: ... of the next sweep
    double* piv = col1 + m + 1;       // n: the pivots
    double* al = piv + n;         // n: alpha
    double* red = al + n;         // THREADS / 64
    const int t = threadIdx.x;
    const int pairs = m * (m + 1) / 2;
    const int cnt = (pairs + THREADS - 1) / THREADS;      // <= MAXE (checked by the launcher)

    double v[MAXE];
    int ij[MAXE];      // (i << 16) | j; the slots past the end of the matrix point at the padding entry m of the column buffers
    if (t == 0) col0[m] = col1[m] = 0.0;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        v[q] = 0.0;
        ij[q] = (m << 16) | m;
        const int idx = q * THREADS + t;
        if (q < cnt && idx < pairs) {
            int i = (int)((__builtin_sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
            while (ltri_i(i + 1) <= idx) ++i;
            while (ltri_i(i) > idx) --i;
            const int j = idx - ltri_i(i);
            ij[q] = (i << 16) | j;
            if (i < n)
                v[q] = os * (gram ? e[(int64_t)i * n + j] : exp(-theta * e[(int64_t)i * n + j])) + (i == j ? noise : 0.0);
            else
                v[q] = (j < n) ? y[j] - mean : 0.0;
            if (j == 0) col0[i] = v[q];
        }
    }

    // One sweep: every LDS read of the step is issued before anything waits on one (the loads do not depend on each other), the
    // update itself is branch-free, and the entries of the next pivot's row / column are published as they are produced.
    bool bad = false;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const double* cur = (k & 1) ? col1 : col0;
        double* nxt = (k & 1) ? col0 : col1;
        double ci[MAXE], cj[MAXE];
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            if (q < cnt) {
                ci[q] = cur[ij[q] >> 16];
                cj[q] = cur[ij[q] & 0xffff];
            }
        }
        const double p = cur[k];
        if (!(p > 0.0)) {          // every thread reads the same value: the exit is uniform
            bad = true;
            break;
        }
        const double ip = rcp(p);
        if (t == 0) piv[k] = p;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            if (q < cnt) {
                const int i = ij[q] >> 16, j = ij[q] & 0xffff;
                const bool ik = (i == k), jk = (j == k);
                const double general = __builtin_fma(-ci[q] * ip, cj[q], v[q]);
                double x = (ik || jk) ? v[q] * ip : general;
                x = (ik && jk) ? -ip : x;
                v[q] = x;
                if (i == k + 1)
                    nxt[j] = x;            // row k+1 (columns <= k+1, the pivot included)
                else if (j == k + 1)
                    nxt[i] = x;            // column k+1 below the diagonal
            }
        }
    }
    if (bad) {
        if (t == 0) {
            out[0] = out[1] = out[2] = out[3] = out[4] = 0.0;
            out[5] = 1.0;
        }
        return;
    }

    // border -> alpha (LDS), corner -> -r.alpha
    double quad_part = 0.0;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        if (q < cnt && (ij[q] >> 16) == n) {
            const int j = ij[q] & 0xffff;
            if (j < n)
                al[j] = v[q];
            else
                quad_part = -v[q];
        }
    }
    const double quad = mll_block_sum<THREADS>(quad_part, red);      // (its barriers also publish al and piv)
    double part = 0.0, asum = 0.0;
    for (int i = t; i < n; i += THREADS) {
        part += log(piv[i]);
        asum += al[i];
    }
    const double logdet = mll_block_sum<THREADS>(part, red);
    const double alpha_sum = mll_block_sum<THREADS>(asum, red);

    // traces of W = alpha alpha^T - Ky^-1 against exp(-theta E) and E o exp(-theta E); the matrix block holds -Ky^-1
    double acc_kb = 0.0, acc_e = 0.0, acc_tr = 0.0;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        if (q < cnt && (ij[q] >> 16) < n) {
            const int i = ij[q] >> 16, j = ij[q] & 0xffff;
            const double w = __builtin_fma(al[i], al[j], v[q]);
            const double eij = e[(int64_t)i * n + j];
            const double kb = gram ? eij : exp(-theta * eij);
            if (w_out) {
                w_out[(int64_t)i * n + j] = w;
                w_out[(int64_t)j * n + i] = w;
            }
            const double wgt = (i == j) ? 1.0 : 2.0;
            acc_kb = __builtin_fma(wgt * w, kb, acc_kb);
            acc_e = __builtin_fma(wgt * w, eij * kb, acc_e);
            if (i == j) acc_tr += w;
        }
    }
    acc_kb = mll_block_sum<THREADS>(acc_kb, red);
    acc_e = mll_block_sum<THREADS>(acc_e, red);
    acc_tr = mll_block_sum<THREADS>(acc_tr, red);
    if (t == 0) {
        out[0] = -0.5 * quad - 0.5 * logdet - 0.5 * (double)n * 1.8378770664093453;      // log(2 pi)
        out[1] = gram ? 0.0 : -0.5 * os * acc_e;
        out[2] = 0.5 * acc_kb;
        out[3] = 0.5 * acc_tr;
        out[4] = alpha_sum;
        out[5] = 0.0;
    }
}

template <int THREADS, int MAXE>
static void launch_mll(const double* e, const double* y, int n, double theta, double os, double noise, double mean, double* out,
                       int gram, double* w_out, hipStream_t st) {
    const size_t lds = (size_t)(2 * (n + 2) + 2 * n + THREADS / 64) * sizeof(double);
    hipLaunchKernelGGL((gp_mll_kernel<THREADS, MAXE>), dim3(1), dim3(THREADS), lds, st, e, y, n, theta, os, noise, mean, out,
                       gram, w_out);
}

}  // namespace gabo

static int gp_mll_dispatch(const double* e, const double* y, int64_t n, double theta, double outputscale, double noise, double mean,
                           double* out, int gram, double* w_out, gabo_stream_t stream) {
    if (n < 1 || n > GABO_GP_MLL_MAX_N) return GABO_ERR_DIM;
    if (!e || !y || !out) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t pairs = (n + 1) * (n + 2) / 2;      // packed entries of the bordered matrix
    if (pairs <= 256 * 2)              // n <= 30
        gabo::launch_mll<256, 2>(e, y, (int)n, theta, outputscale, noise, mean, out, gram, w_out, st);
    else if (pairs <= 256 * 8)         // n <= 62
        gabo::launch_mll<256, 8>(e, y, (int)n, theta, outputscale, noise, mean, out, gram, w_out, st);
    else if (pairs <= 512 * 13)        // n <= 113
        gabo::launch_mll<512, 13>(e, y, (int)n, theta, outputscale, noise, mean, out, gram, w_out, st);
    else                               // n <= 160 (GABO_GP_MLL_MAX_N): 26 entries per thread is what 256 VGPRs hold without spilling
        gabo::launch_mll<512, 26>(e, y, (int)n, theta, outputscale, noise, mean, out, gram, w_out, st);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

extern "C" int gabo_gp_mll(const double* e, const double* y, int64_t n, double theta, double outputscale, double noise, double mean,
                           double* out, gabo_stream_t stream) {
    return gp_mll_dispatch(e, y, n, theta, outputscale, noise, mean, out, 0, nullptr, stream);
}

extern "C" int gabo_gp_mll_gram(const double* k, const double* y, int64_t n, double outputscale, double noise, double mean, double* out,
                                double* w, gabo_stream_t stream) {
    return gp_mll_dispatch(k, y, n, 0.0, outputscale, noise, mean, out, 1, w, stream);
}
