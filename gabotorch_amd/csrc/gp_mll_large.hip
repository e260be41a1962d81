// Exact-GP marginal log likelihood and its analytic gradient for training sets beyond the single-workgroup kernel of gp_mll.hip
// (GABO_GP_MLL_MAX_N = 160 < n <= GABO_GP_MLL_LARGE_MAX_N): the same symmetric sweep operator, blocked.  Same contract as gabo_gp_mll /
// gabo_gp_mll_gram (see gp_mll.hip for the formulas and the reference call site: `fit_gpytorch_model(mll)`, examples/.../gabo_spd.py:194).
//
// Without this the surrogate fit fell off a cliff at n = 161 (9.6 ms at n = 160, 56 ms at n = 200, 338 ms at n = 1024: autograd through the
// pairwise kernel, torch's Cholesky and solves on every L-BFGS evaluation).
//
// The bordered matrix [[Ky, r], [r^T, 0]] lives in the caller's workspace as b x b tiles (b = 32; Ky padded with an identity block to a
// multiple of b, the border in a block row of its own).  Sweeping on a whole pivot BLOCK P is the block form of the scalar sweep,
//     P <- -P^-1,   B <- B P^-1 (every other block row),   C <- C - B P^-1 B^T (every other pair of block rows),
// and sweeping the blocks of Ky one after the other leaves -Ky^-1 in the matrix part, alpha = Ky^-1 r in the border and -r.alpha in the corner;
// det Ky is the product of the determinants of the pivot blocks as they are met (Schur complements: positive definite iff Ky is).
// One step is ONE launch, embarrassingly parallel over tiles (mll_step_kernel: one block per tile (i, j)):
//   the sweep on pivot block k:   M_ij -= G_i H_j^T,  M_ik <- G_i,  M_kj <- G_j^T,  M_kk <- -P^-1   with the panel G_i = M_ik P^-1, H_i = M_ik;
//   the blocks of tile column k + 1 go on to produce the NEXT panel: each rebuilds the updated pivot tile (k+1, k+1) itself, inverts it by 32
//   scalar sweeps in LDS (repeated by every block of the column, which costs nothing on an otherwise idle chip and saves a launch and a
//   wait per step) and forms G_i(k+1), H_i(k+1) into the second buffer set
// then one pass over the tiles for the traces of W = alpha alpha^T - Ky^-1 against exp(-theta E) and E o exp(-theta E), reduced in a fixed order.
// n^3 FMAs in all (both triangles are carried: the tiles stay plain dense products), n / 32 + 4 launches: launch-latency-bound below n ~ 500.
#include "gabo_device.hpp"
#include "lds_linalg.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

constexpr int kMllB = 32;                  // tile order

struct MllLargeLayout {
    int n, nr, nt, np;                     // training points; Ky padded to nr = ceil(n / b) b; block rows nt = nr / b + 1 (the border's); np = nt b
    double *M, *G, *H, *Dg, *acc, *part;   // np x np tiles (tile-major: tile (I, J) at ((I nt + J) b b)); two sets of np x b each (G, H: step k reads
                                           // set k & 1 and writes the other); two sets of the nt diagonal tiles; 8 doubles; partial sums
    size_t bytes;
    __host__ __device__ double* g(int set, int i) const { return G + ((size_t)set * nt + i) * kMllB * kMllB; }
    __host__ __device__ double* h(int set, int i) const { return H + ((size_t)set * nt + i) * kMllB * kMllB; }
    __host__ __device__ double* dg(int set, int i) const { return Dg + ((size_t)set * nt + i) * kMllB * kMllB; }
};

static __host__ __device__ inline MllLargeLayout mll_large_layout(void* base, int n) {
    MllLargeLayout L;
    L.n = n;
    L.nr = (n + kMllB - 1) / kMllB * kMllB;
    L.nt = L.nr / kMllB + 1;
    L.np = L.nt * kMllB;
    double* p = static_cast<double*>(base);
    L.M = p;      p += (size_t)L.np * L.np;
    L.G = p;      p += (size_t)2 * L.np * kMllB;
    L.H = p;      p += (size_t)2 * L.np * kMllB;
    L.Dg = p;     p += (size_t)2 * L.np * kMllB;
    L.acc = p;    p += 8;                  // [0] log det so far, [1] not-positive-definite flag (as a double)
    L.part = p;   p += (size_t)4 * L.nt * L.nt;
    L.bytes = (size_t)((char*)p - (char*)base);
    return L;
}

// tile (I, J) of the bordered matrix, entry (a, b): global row I b + a.  Rows / columns < n: Ky; n .. nr-1: identity padding; nr: the border
// (r = y - mean), the corner 0; beyond: identity padding.
__global__ __launch_bounds__(256) void mll_build_kernel(const double* __restrict__ e, const double* __restrict__ y, MllLargeLayout L, double theta,
                                                        double os, double noise, double mean, int gram) {
    const int I = blockIdx.x / L.nt, J = blockIdx.x - I * L.nt;
    double* T = L.M + ((size_t)I * L.nt + J) * kMllB * kMllB;
    for (int q = threadIdx.x; q < kMllB * kMllB; q += blockDim.x) {
        const int a = q / kMllB, b = q - a * kMllB;
        const int i = I * kMllB + a, j = J * kMllB + b;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < L.n && j < L.n) {
            const double x = e[(size_t)i * L.n + j];
            v = os * (gram ? x : exp(-theta * x)) + (i == j ? noise : 0.0);
        } else if (i == L.nr && j < L.n) {
            v = y[j] - mean;
        } else if (j == L.nr && i < L.n) {
            v = y[i] - mean;
        } else if (i == L.nr && j == L.nr) {
            v = 0.0;
        }
        T[q] = v;
        if (I == J) L.dg(0, I)[q] = v;            // (the copy the first step reads: see mll_step_kernel)
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) L.acc[threadIdx.x] = 0.0;
}

// -P^-1 of the 32 x 32 pivot block in LDS by 32 scalar sweeps (the operator of gp_mll.hip on one tile: every entry updated in parallel,
// two barriers per pivot); returns (uniformly) whether every pivot was positive, *logdet = sum log pivot.  256 threads, 4 entries each.
static __device__ __forceinline__ bool sweep_tile(double* P, double* col, double* logdet) {
    const int t = threadIdx.x;
    bool ok = true;
    double mine = 1.0;                                      // thread k keeps pivot k: the logarithms are taken once, after the sweeps
    for (int k = 0; k < kMllB; ++k) {
        if (t < kMllB) col[t] = P[t * kMllB + k];          // column k (= row k: the tile is symmetric) before it is rewritten
        __syncthreads();
        const double p = col[k];
        ok = ok && (p > 0.0);
        mine = (t == k) ? p : mine;
        const double ip = rcp(p);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = t + u * 256;
            const int a = q / kMllB, b = q - a * kMllB;
            const double ca = col[a], cb = col[b];
            double v;
            if (a == k && b == k) v = -ip;
            else if (a == k) v = cb * ip;
            else if (b == k) v = ca * ip;
            else v = __builtin_fma(-ca * ip, cb, P[q]);
            P[q] = v;
        }
        __syncthreads();
    }
    // log det = sum of the logs of the 32 pivots (threads 0..31 = the first half of wave 0), the same value in every thread through LDS
    double ld = (t < kMllB) ? log(mine) : 0.0;
    if (t < 64) {
        for (int off = 32; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
        if (t == 0) col[0] = ld;
    }
    __syncthreads();
    *logdet = col[0];
    __syncthreads();
    return ok;
}

// The panel of the FIRST pivot block (k = 0), block row i: P = M_00 -> -P^-1 (every block; block 0 also records log det P and the
// definiteness flag), G_i = M_i0 P^-1, H_i = M_i0, into set 0.  Later panels are produced by the step kernel of the pivot before.
__global__ __launch_bounds__(256) void mll_panel_kernel(MllLargeLayout L) {
    __shared__ __attribute__((aligned(16))) double P[kMllB * kMllB], T[kMllB * kMllB], col[kMllB];
    const int i = blockIdx.x;
    constexpr int bb = kMllB * kMllB;
    const double* Pk = L.M;
    for (int q = threadIdx.x; q < bb; q += blockDim.x) P[q] = Pk[q];
    if (i != 0) {
        const double* Ti = L.M + ((size_t)i * L.nt) * bb;
        for (int q = threadIdx.x; q < bb; q += blockDim.x) T[q] = Ti[q];
    }
    __syncthreads();
    double logdet;
    const bool ok = sweep_tile(P, col, &logdet);            // P = -P^-1
    if (i == 0) {
        double* out = L.g(0, 0);
        for (int q = threadIdx.x; q < bb; q += blockDim.x) out[q] = P[q];
        if (threadIdx.x == 0) {
            L.acc[0] += logdet;
            if (!ok) L.acc[1] = 1.0;
        }
        return;
    }
    double* Gi = L.g(0, i);
    double* Hi = L.h(0, i);
    for (int q = threadIdx.x; q < bb; q += blockDim.x) {
        const int a = q / kMllB, b = q - a * kMllB;
        double s = 0.0;
        for (int c = 0; c < kMllB; ++c) s = __builtin_fma(T[a * kMllB + c], P[c * kMllB + b], s);
        Gi[q] = -s;                                         // M_i0 P^-1
        Hi[q] = T[q];
    }
}

// out (LDS, 32 x 32) = base - Ga Hb^T, or = base when Ga == nullptr.  A, B: LDS staging (B padded: Hb^T is read down a column).
static __device__ __forceinline__ void tile_update(const double* __restrict__ base, const double* __restrict__ Ga, const double* __restrict__ Hb,
                                                   double* A, double* B, double* out) {
    constexpr int bb = kMllB * kMllB;
    for (int q = threadIdx.x; q < bb; q += blockDim.x) {
        A[q] = Ga[q];
        B[(q / kMllB) * (kMllB + 1) + (q % kMllB)] = Hb[q];
    }
    __syncthreads();
    const int ta = (threadIdx.x >> 4) * 2, tb = (threadIdx.x & 15) * 2;      // thread -> a 2 x 2 patch of the tile (256 threads x 4 outputs)
    double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
#pragma unroll 8
    for (int c = 0; c < kMllB; ++c) {
        const double a0 = A[ta * kMllB + c], a1 = A[(ta + 1) * kMllB + c];
        const double b0 = B[tb * (kMllB + 1) + c], b1 = B[(tb + 1) * (kMllB + 1) + c];
        s00 = __builtin_fma(a0, b0, s00);
        s01 = __builtin_fma(a0, b1, s01);
        s10 = __builtin_fma(a1, b0, s10);
        s11 = __builtin_fma(a1, b1, s11);
    }
    out[ta * kMllB + tb] = base[ta * kMllB + tb] - s00;
    out[ta * kMllB + tb + 1] = base[ta * kMllB + tb + 1] - s01;
    out[(ta + 1) * kMllB + tb] = base[(ta + 1) * kMllB + tb] - s10;
    out[(ta + 1) * kMllB + tb + 1] = base[(ta + 1) * kMllB + tb + 1] - s11;
    __syncthreads();
}

// Step k, tile (i, j): the sweep on pivot block k applied to the tile -  M_ij -= G_i H_j^T,  M_ik <- G_i,  M_kj <- G_j^T,  M_kk <- -P^-1
// (G, H of set k & 1) - and, in the tile column of the NEXT pivot (j = k + 1, when has_next), that pivot's panel into the other set: the block
// rebuilds the updated pivot tile itself (from the diagonal copy of the previous step: the tile in M is being rewritten by another block of
// this launch), inverts it by 32 scalar sweeps, and forms G_i(k+1) = M'_i,k+1 P^-1, H_i(k+1) = M'_i,k+1.  One launch per pivot block.
__global__ __launch_bounds__(256) void mll_step_kernel(MllLargeLayout L, int k, int has_next) {
    __shared__ __attribute__((aligned(16))) double A[kMllB * kMllB], B[kMllB * (kMllB + 1)], Tn[kMllB * kMllB], P[kMllB * kMllB], col[kMllB];
    constexpr int bb = kMllB * kMllB;
    const int i = blockIdx.x / L.nt, j = blockIdx.x - i * L.nt;
    const int cur = k & 1, nxt = cur ^ 1;
    double* Tij = L.M + ((size_t)i * L.nt + j) * bb;
    if (i == k && j == k) {
        const double* negPinv = L.g(cur, k);
        for (int q = threadIdx.x; q < bb; q += blockDim.x) Tn[q] = negPinv[q];
        __syncthreads();
    } else if (j == k) {
        const double* Gi = L.g(cur, i);
        for (int q = threadIdx.x; q < bb; q += blockDim.x) Tn[q] = Gi[q];
        __syncthreads();
    } else if (i == k) {
        const double* Gj = L.g(cur, j);
        for (int q = threadIdx.x; q < bb; q += blockDim.x) {
            const int a = q / kMllB, b = q - a * kMllB;
            Tn[q] = Gj[b * kMllB + a];
        }
        __syncthreads();
    } else {
        tile_update(Tij, L.g(cur, i), L.h(cur, j), A, B, Tn);
    }
    for (int q = threadIdx.x; q < bb; q += blockDim.x) {
        Tij[q] = Tn[q];
        if (i == j) L.dg(nxt, i)[q] = Tn[q];                // the diagonal copy the next step's panel blocks read
    }
    if (!has_next || j != k + 1) return;
    // ---- panel of pivot k + 1
    const int kn = k + 1;
    if (i == kn) {
        for (int q = threadIdx.x; q < bb; q += blockDim.x) P[q] = Tn[q];         // this IS the updated pivot tile
        __syncthreads();
    } else {
        tile_update(L.dg(cur, kn), L.g(cur, kn), L.h(cur, kn), A, B, P);           // (kn != k: the general update of tile (kn, kn))
    }
    double logdet;
    const bool ok = sweep_tile(P, col, &logdet);            // P = -P^-1
    if (i == kn) {
        double* out = L.g(nxt, kn);
        for (int q = threadIdx.x; q < bb; q += blockDim.x) out[q] = P[q];
        if (threadIdx.x == 0) {
            L.acc[0] += logdet;                             // (one writer per step, steps are stream-ordered)
            if (!ok) L.acc[1] = 1.0;
        }
        return;
    }
    double* Gi = L.g(nxt, i);
    double* Hi = L.h(nxt, i);
    for (int q = threadIdx.x; q < bb; q += blockDim.x) {
        const int a = q / kMllB, b = q - a * kMllB;
        double s = 0.0;
        for (int c = 0; c < kMllB; ++c) s = __builtin_fma(Tn[a * kMllB + c], P[c * kMllB + b], s);
        Gi[q] = -s;
        Hi[q] = Tn[q];
    }
}

// traces of W = alpha alpha^T - Ky^-1 over tile (I, J) of the Ky part: partial sums [W.kb, W.(E o kb), tr W, sum alpha (J == 0 tiles)]
__global__ __launch_bounds__(256) void mll_trace_kernel(const double* __restrict__ e, MllLargeLayout L, double theta, int gram,
                                                        double* __restrict__ w_out) {
    __shared__ double red[4][4];
    constexpr int bb = kMllB * kMllB;
    const int nb = L.nr / kMllB;
    const int I = blockIdx.x / nb, J = blockIdx.x - I * nb;
    const double* T = L.M + ((size_t)I * L.nt + J) * bb;
    const double* al = L.M + ((size_t)(L.nt - 1) * L.nt) * bb;      // border block row: tile (nt - 1, J'), row 0 = alpha
    double s_kb = 0.0, s_e = 0.0, s_tr = 0.0, s_al = 0.0;
    for (int q = threadIdx.x; q < bb; q += blockDim.x) {
        const int a = q / kMllB, b = q - a * kMllB;
        const int i = I * kMllB + a, j = J * kMllB + b;
        if (i < L.n && j < L.n) {
            const double ai = al[(size_t)I * bb + a], aj = al[(size_t)J * bb + b];
            const double w = __builtin_fma(ai, aj, T[q]);
            const double eij = e[(size_t)i * L.n + j];
            const double kb = gram ? eij : exp(-theta * eij);
            if (w_out) w_out[(size_t)i * L.n + j] = w;
            s_kb = __builtin_fma(w, kb, s_kb);
            s_e = __builtin_fma(w, eij * kb, s_e);
            if (i == j) s_tr += w;
            if (J == 0 && b == 0) s_al += ai;
        }
    }
    double v[4] = {s_kb, s_e, s_tr, s_al};
    for (int c = 0; c < 4; ++c) {
        double x = v[c];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) L.part[(size_t)blockIdx.x * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void mll_final_kernel(MllLargeLayout L, double os, int gram, double* __restrict__ out) {
    __shared__ double red[4][4];
    const int nb = L.nr / kMllB, tiles = nb * nb;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = threadIdx.x; t < tiles; t += blockDim.x)
        for (int c = 0; c < 4; ++c) v[c] += L.part[(size_t)t * 4 + c];
    for (int c = 0; c < 4; ++c) {
        double x = v[c];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[4];
        for (int c = 0; c < 4; ++c) s[c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
        const double corner = L.M[((size_t)(L.nt - 1) * L.nt + (L.nt - 1)) * kMllB * kMllB];      // -r.alpha
        const bool bad = L.acc[1] != 0.0 || !(L.acc[0] == L.acc[0]);
        if (bad) {
            out[0] = out[1] = out[2] = out[3] = out[4] = 0.0;
            out[5] = 1.0;
        } else {
            out[0] = 0.5 * corner - 0.5 * L.acc[0] - 0.5 * (double)L.n * 1.8378770664093453;      // log(2 pi)
            out[1] = gram ? 0.0 : -0.5 * os * s[1];
            out[2] = 0.5 * s[0];
            out[3] = 0.5 * s[2];
            out[4] = s[3];
            out[5] = 0.0;
        }
    }
}

}  // namespace gabo

extern "C" {

size_t gabo_gp_mll_large_workspace_bytes(int64_t n) {
    if (n < 1 || n > GABO_GP_MLL_LARGE_MAX_N) return 0;
    return gabo::mll_large_layout(nullptr, (int)n).bytes;
}

int gabo_gp_mll_large(const double* e, const double* y, int64_t n, double theta, double outputscale, double noise, double mean, int gram,
                      double* out, double* w, void* workspace, size_t workspace_bytes, gabo_stream_t stream) {
    if (n < 1 || n > GABO_GP_MLL_LARGE_MAX_N) return GABO_ERR_DIM;
    if (!e || !y || !out || !workspace || workspace_bytes < gabo_gp_mll_large_workspace_bytes(n)) return GABO_ERR_ARG;
    const gabo::MllLargeLayout L = gabo::mll_large_layout(workspace, (int)n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gabo::mll_build_kernel, dim3((unsigned)(L.nt * L.nt)), dim3(256), 0, st, e, y, L, theta, outputscale, noise, mean, gram);
    hipLaunchKernelGGL(gabo::mll_panel_kernel, dim3((unsigned)L.nt), dim3(256), 0, st, L);
    for (int k = 0; k < L.nt - 1; ++k)
        hipLaunchKernelGGL(gabo::mll_step_kernel, dim3((unsigned)(L.nt * L.nt)), dim3(256), 0, st, L, k, (k + 1 < L.nt - 1) ? 1 : 0);
    const int nb = L.nr / gabo::kMllB;
    hipLaunchKernelGGL(gabo::mll_trace_kernel, dim3((unsigned)(nb * nb)), dim3(256), 0, st, e, L, theta, gram, w);
    hipLaunchKernelGGL(gabo::mll_final_kernel, dim3(1), dim3(256), 0, st, L, outputscale, gram, out);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
}
