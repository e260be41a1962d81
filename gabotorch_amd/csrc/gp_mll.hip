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
// [[Ky, r], [r^T, 0]] distributed over the threads' REGISTERS as 4 x 4 tiles of its lower triangle (one tile per thread, diagonal
// tiles stored in full), and the symmetric sweep operator (Gauss-Jordan on the pivots 0..n-1: a_kk <- -1/p, a_ik <- a_ik/p,
// a_ij <- a_ij - a_ik a_jk/p) applied in place.  After the n sweeps the matrix block holds -Ky^-1, the border holds alpha and the
// corner -r.alpha; the pivots are the Schur complements, so log det(Ky) = sum log p_k and Ky is positive definite iff every p_k > 0.
// One sweep is one barrier: only the pivot column travels through LDS (double-buffered).  A thread reads the 4 + 4 column entries of
// its tile's rows and columns and does 16 FMAs, x_ab <- x_ab - g_a h_b with g = c_row / p, h = c_col.  The pivot row and column need no
// special case in that update: their owners publish them for the next sweep one step ahead and ZERO their register copies, and the
// sweep uses g = -1/p at the pivot row and h = -1 at the pivot column, which turns the same FMA into c_i / p, c_j / p and -1/p there.
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

#ifndef GABO_MLL_SMALL_N
#define GABO_MLL_SMALL_N 30
#endif

namespace gabo {

constexpr int kMllMaxThreads = 1024;

static __device__ __forceinline__ double mll_block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;
}

// publish row R / column R of the tile as entries of the next pivot's column, then zero them (see the header comment)
template <int R>
static __device__ __forceinline__ void mll_publish(double (&x)[4][4], int I, int J, int kt1, double* __restrict__ nxt) {
    if (J == kt1) {                       // the tile holds column k+1 (rows 4I .. 4I+3, the diagonal tile included)
        static_for<4>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            nxt[4 * I + a] = x[a][R];
            x[a][R] = 0.0;
        });
    }
    if (I == kt1) {                       // the tile holds row k+1; columns left of the diagonal tile are published from here
        static_for<4>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            if (J != kt1) nxt[4 * J + b] = x[R][b];
            x[R][b] = 0.0;
        });
    }
}

// blockDim.x = number of tiles rounded up to whole waves; tile (I, J), I >= J, of thread t in row-major packed order
// MAXT = 64: the whole matrix fits one wave's tiles (n <= 39) and the compiler drops the barriers of a single-wave workgroup
template <int MAXT>
__global__ __launch_bounds__(MAXT) void gp_mll_kernel(const double* __restrict__ e, const double* __restrict__ y, int n,
                                                               double theta, double os, double noise, double mean,
                                                               double* __restrict__ out, int gram, double* __restrict__ w_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = n + 1;                       // bordered matrix
    const int nt = (m + 3) / 4, mp = 4 * nt;   // tiles per side, padded size
    double* col0 = lds;            // mp: the pivot column of the current sweep (entry k is the pivot itself)
    double* col1 = col0 + mp;      // mp: ... of the next sweep
    double* piv = col1 + mp;       // n
    double* al = piv + n;          // mp: alpha
    double* red = al + mp;         // 16
    const int t = threadIdx.x;
    const int tiles = nt * (nt + 1) / 2;
    const bool live = t < tiles;
    int I = 0, J = 0;
    if (live) {
        I = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        while (I * (I + 1) / 2 > t) --I;
        J = t - I * (I + 1) / 2;
    }
    // every global load of the tile is issued before anything waits on one (clamped addresses instead of branches)
    double x[4][4], ev[4][4], yv[4];
    static_for<4>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        const int j = 4 * J + b;
        yv[b] = y[j < n ? j : n - 1];
    });
    static_for<4>([&](auto aa) {
        constexpr int a = decltype(aa)::value;
        static_for<4>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            const int i = 4 * I + a, j = 4 * J + b;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            ev[a][b] = e[hi < n ? (int64_t)hi * n + lo : 0];
        });
    });
    static_for<4>([&](auto aa) {
        constexpr int a = decltype(aa)::value;
        static_for<4>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            const int i = 4 * I + a, j = 4 * J + b;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            const double kern = os * (gram ? ev[a][b] : exp(-theta * ev[a][b])) + (i == j ? noise : 0.0);
            // border row (hi == n; only lower tiles own it, so lo = j), corner, identity padding
            const double ident = (i == j) ? 1.0 : 0.0;
            const double other = (hi == n) ? ((lo < n) ? yv[i > j ? b : a] - mean : 0.0) : ident;      // (diagonal tiles: I == J)
            x[a][b] = !live ? ident : (hi < n ? kern : other);
        });
    });
    // column 0 for the first sweep, published and zeroed like every later one
    if (live) mll_publish<0>(x, I, J, 0, col0);

    bool bad = false;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const double* cur = (k & 1) ? col1 : col0;
        double* nxt = (k & 1) ? col0 : col1;
        const int kt = k >> 2, kr = k & 3;
        double ci[4], cj[4];
        static_for<4>([&](auto aa) {
            ci[decltype(aa)::value] = cur[4 * I + decltype(aa)::value];
            cj[decltype(aa)::value] = cur[4 * J + decltype(aa)::value];
        });
        const double p = cur[k];
        if (!(p > 0.0)) {          // every thread reads the same value: the exit is uniform
            bad = true;
            break;
        }
        const double ip = rcp(p);
        if (t == 0) piv[k] = p;
        double g[4], h[4];
        static_for<4>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            g[a] = (I == kt && a == kr) ? -ip : ci[a] * ip;
            h[a] = (J == kt && a == kr) ? -1.0 : cj[a];
        });
        static_for<4>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            static_for<4>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                x[a][b] = __builtin_fma(-g[a], h[b], x[a][b]);
            });
        });
        if (live) {
            const int k1 = k + 1, kt1 = k1 >> 2;
            switch (k1 & 3) {          // wave-uniform
                case 0: mll_publish<0>(x, I, J, kt1, nxt); break;
                case 1: mll_publish<1>(x, I, J, kt1, nxt); break;
                case 2: mll_publish<2>(x, I, J, kt1, nxt); break;
                default: mll_publish<3>(x, I, J, kt1, nxt); break;
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
    __syncthreads();
    // The border row n was published as "row k+1" at the last sweep and zeroed in the registers: alpha is that column buffer
    const double* fin = (n & 1) ? col1 : col0;
    for (int i = t; i < mp; i += blockDim.x) al[i] = (i < n) ? fin[i] : 0.0;
    const double quad = -fin[n];                  // corner: -r.alpha (every thread reads the same LDS word)
    double part = 0.0, asum = 0.0;
    __syncthreads();
    for (int i = t; i < n; i += blockDim.x) {
        part += log(piv[i]);
        asum += al[i];
    }
    const double logdet = mll_block_sum(part, red);
    const double alpha_sum = mll_block_sum(asum, red);

    // traces of W = alpha alpha^T - Ky^-1 against exp(-theta E) and E o exp(-theta E); the matrix block holds -Ky^-1
    double acc_kb = 0.0, acc_e = 0.0, acc_tr = 0.0;
    static_for<4>([&](auto aa) {               // (reloaded rather than kept: 32 registers less during the sweeps)
        constexpr int a = decltype(aa)::value;
        static_for<4>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            const int i = 4 * I + a, j = 4 * J + b;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            ev[a][b] = e[hi < n ? (int64_t)hi * n + lo : 0];
        });
    });
    if (live) {
        static_for<4>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            static_for<4>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                const int i = 4 * I + a, j = 4 * J + b;
                if (i < n && j < n && j <= i) {           // every unordered pair once (diagonal tiles hold both triangles)
                    const double w = __builtin_fma(al[i], al[j], x[a][b]);
                    const double eij = ev[a][b];
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
            });
        });
    }
    acc_kb = mll_block_sum(acc_kb, red);
    acc_e = mll_block_sum(acc_e, red);
    acc_tr = mll_block_sum(acc_tr, red);
    if (t == 0) {
        out[0] = -0.5 * quad - 0.5 * logdet - 0.5 * (double)n * 1.8378770664093453;      // log(2 pi)
        out[1] = gram ? 0.0 : -0.5 * os * acc_e;
        out[2] = 0.5 * acc_kb;
        out[3] = 0.5 * acc_tr;
        out[4] = alpha_sum;
        out[5] = 0.0;
    }
}

static __device__ __forceinline__ int ltri_i(int r) { return r * (r + 1) / 2; }

template <int THREADS>
static __device__ __forceinline__ double mll_small_block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < THREADS / 64; ++k) s += red[k];
    return s;
}

// Small training sets (n <= 30): the same sweeps with the packed lower triangle distributed entry by entry (MAXE entries per thread,
// entry idx = q * THREADS + t) - the 4 x 4 tiles above would spend more on their 32 exponentials per thread than on the sweeps.
template <int THREADS, int MAXE>
__global__ __launch_bounds__(THREADS) void gp_mll_small_kernel(const double* __restrict__ e, const double* __restrict__ y, int n,
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
    const double quad = mll_small_block_sum<THREADS>(quad_part, red);      // (its barriers also publish al and piv)
    double part = 0.0, asum = 0.0;
    for (int i = t; i < n; i += THREADS) {
        part += log(piv[i]);
        asum += al[i];
    }
    const double logdet = mll_small_block_sum<THREADS>(part, red);
    const double alpha_sum = mll_small_block_sum<THREADS>(asum, red);

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
    acc_kb = mll_small_block_sum<THREADS>(acc_kb, red);
    acc_e = mll_small_block_sum<THREADS>(acc_e, red);
    acc_tr = mll_small_block_sum<THREADS>(acc_tr, red);
    if (t == 0) {
        out[0] = -0.5 * quad - 0.5 * logdet - 0.5 * (double)n * 1.8378770664093453;      // log(2 pi)
        out[1] = gram ? 0.0 : -0.5 * os * acc_e;
        out[2] = 0.5 * acc_kb;
        out[3] = 0.5 * acc_tr;
        out[4] = alpha_sum;
        out[5] = 0.0;
    }
}


}  // namespace gabo

static int gp_mll_dispatch(const double* e, const double* y, int64_t n, double theta, double outputscale, double noise, double mean,
                           double* out, int gram, double* w_out, gabo_stream_t stream) {
    if (n < 1 || n > GABO_GP_MLL_MAX_N) return GABO_ERR_DIM;
    if (!e || !y || !out) return GABO_ERR_ARG;
    if (n <= GABO_MLL_SMALL_N) {                          // (n + 1)(n + 2) / 2 <= 512 packed entries: two per thread
        const size_t lds_small = (size_t)(2 * (n + 2) + 2 * n + 4) * sizeof(double);
        hipLaunchKernelGGL((gabo::gp_mll_small_kernel<256, 2>), dim3(1), dim3(256), lds_small, (hipStream_t)stream, e, y, (int)n, theta,
                           outputscale, noise, mean, out, gram, w_out);
        return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
    }
    const int nt = (int)((n + 1 + 3) / 4);
    const int tiles = nt * (nt + 1) / 2;                   // <= 861 at n = 160
    const int threads = ((tiles + 63) / 64) * 64;
    const size_t lds = (size_t)(3 * 4 * nt + n + 16) * sizeof(double);
    if (threads == 64)
        hipLaunchKernelGGL(gabo::gp_mll_kernel<64>, dim3(1), dim3(64), lds, (hipStream_t)stream, e, y, (int)n, theta, outputscale, noise,
                           mean, out, gram, w_out);
    else
        hipLaunchKernelGGL(gabo::gp_mll_kernel<gabo::kMllMaxThreads>, dim3(1), dim3(threads), lds, (hipStream_t)stream, e, y, (int)n,
                           theta, outputscale, noise, mean, out, gram, w_out);
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
