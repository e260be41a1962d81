// Small dense linear algebra on d x d tiles held in LDS, executed cooperatively by the threads of one block (one wave):
// the building blocks of the wave-per-matrix kernels (spd_manifold.hip) and of the wave-per-pair fallback for d > 12
// (spd_pairwise_generic.hip).  Every function is called by all threads of the block and ends with a barrier.
#pragma once
#include "gabo_device.hpp"
#include "wave_eigh.hpp"

namespace gabo {

enum { FN_LOG = 0, FN_EXP = 1, FN_SQRT = 2 };

static __device__ __forceinline__ void wsync() { __syncthreads(); }

// symmetric matrix from a Mandel vector (global) into LDS
static __device__ void lds_from_mandel(const double* __restrict__ v, double* A, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        int hi = r > c ? r : c, lo = r > c ? c : r;
        double x = v[mandel_pos(d, hi, lo)];
        A[e] = (r == c) ? x : x * kInvSqrt2;
    }
    wsync();
}

static __device__ void lds_load(const double* __restrict__ src, double* A, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) A[e] = src[e];
    wsync();
}

// A <- (A + A^T)/2 using T as scratch
static __device__ void lds_symmetrize(double* A, double* T, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        T[e] = 0.5 * (A[e] + A[c * d + r]);
    }
    wsync();
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) A[e] = T[e];
    wsync();
}

// in-place lower Cholesky (strict upper zeroed).  Returns false when a pivot is not positive.
static __device__ bool lds_cholesky(double* A, int d) {
    bool ok = true;
    for (int c = 0; c < d; ++c) {
        double piv = A[c * d + c];
        for (int k = 0; k < c; ++k) piv -= A[c * d + k] * A[c * d + k];
        if (!(piv > 0.0)) ok = false;
        const double inv = rsqrt_nz(piv);           // (see mandel_cholesky, spd_prep.hpp)
        const double lcc = piv * inv;
        wsync();
        for (int r = c + threadIdx.x; r < d; r += blockDim.x) {
            if (r == c) {
                A[c * d + c] = lcc;
            } else {
                double s = A[r * d + c];
                for (int k = 0; k < c; ++k) s -= A[r * d + k] * A[c * d + k];
                A[r * d + c] = s * inv;
            }
        }
        for (int r = threadIdx.x; r < c; r += blockDim.x) A[r * d + c] = 0.0;
        wsync();
    }
    return ok;
}

// W = L^-1 for lower-triangular L: thread c owns column c (forward substitution)
static __device__ void lds_tri_inverse(const double* L, double* W, int d) {
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        for (int r = 0; r < c; ++r) W[r * d + c] = 0.0;
        W[c * d + c] = rcp(L[c * d + c]);
        for (int r = c + 1; r < d; ++r) {
            double s = 0.0;
            for (int k = c; k < r; ++k) s += L[r * d + k] * W[k * d + c];
            W[r * d + c] = -s * rcp(L[r * d + r]);
        }
    }
    wsync();
}

// C = op(A) op(B), thread per output element; C must not alias A or B
// (round 4: four products in flight per thread, the transposition flags resolved outside the loop.  Worth ~7 % of a product phase only
// (tools/recon_clocks.py: 17.1 k -> 15.9 k cycles for the five products of "C^1/2, B, Xrec" at D = 20): a 20 x 20 product on 256 threads is
// 320 ds_read_b64 wave instructions through one CU's LDS pipe, ~1.3 k cycles of bandwidth before any latency - a 2 x 2 register tile per
// thread would halve that)
template <bool TA, bool TB>
static __device__ __forceinline__ void lds_mm_body(const double* A, const double* B, double* C, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        const int r = e / d, c = e - r * d;
        const double* ap = TA ? A + r : A + r * d;
        const double* bp = TB ? B + c * d : B + c;
        const int as = TA ? d : 1, bs = TB ? 1 : d;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
        for (; k + 4 <= d; k += 4) {
            const double a0 = ap[k * as], a1 = ap[(k + 1) * as], a2 = ap[(k + 2) * as], a3 = ap[(k + 3) * as];
            const double b0 = bp[k * bs], b1 = bp[(k + 1) * bs], b2 = bp[(k + 2) * bs], b3 = bp[(k + 3) * bs];
            s0 = __builtin_fma(a0, b0, s0);
            s1 = __builtin_fma(a1, b1, s1);
            s2 = __builtin_fma(a2, b2, s2);
            s3 = __builtin_fma(a3, b3, s3);
        }
        for (; k < d; ++k) s0 = __builtin_fma(ap[k * as], bp[k * bs], s0);
        C[e] = (s0 + s1) + (s2 + s3);
    }
}
static __device__ void lds_mm(const double* A, const double* B, double* C, int d, bool ta, bool tb) {
    if (ta) {
        if (tb) lds_mm_body<true, true>(A, B, C, d);
        else lds_mm_body<true, false>(A, B, C, d);
    } else {
        if (tb) lds_mm_body<false, true>(A, B, C, d);
        else lds_mm_body<false, false>(A, B, C, d);
    }
    wsync();
}

// C = A B A^T (congruence), T scratch
static __device__ void lds_congruence(const double* A, const double* B, double* C, double* T, int d) {
    lds_mm(A, B, T, d, false, false);
    lds_mm(T, A, C, d, false, true);
}

// LDS scratch (doubles) the eigen-solvers need next to the matrices: lds_jacobi keeps (c, s) and the index pair of up to 16 concurrent
// rotations there (48 doubles), wave_eigh the (u, q) pairs of a Householder step (kWaveEighScratch = 64)
constexpr int kJacobiScratch = kWaveEighScratch;

// pair (p < q) of slot s in round r of the circle-method schedule over np + 1 players (np odd); q may be the padding player
static __device__ __forceinline__ void jacobi_pair(int r, int s, int np, int& p, int& q) {
    const int a = (s == 0) ? np : (r + s) % np;
    const int b = (s == 0) ? r : (r - s + np) % np;
    p = a < b ? a : b;
    q = a < b ? b : a;
}

// Jacobi eigen-decomposition: A (symmetric, full storage) -> diagonal; V = eigenvectors in columns (V may be null: eigenvalues
// only).  cs: kJacobiScratch doubles of LDS.  The block is one wave, or up to four (blockDim.x a multiple of 64, <= 256): above
// d = 12 a round's 2 x d x d/2 work items are worth spreading over 256 threads.
// Parallel ordering: a sweep is d-1 (d even) or d (d odd) rounds of floor(d/2) rotations on disjoint index pairs (round-robin
// tournament), so a round costs three barrier phases - angles, columns of all pairs, rows of all pairs - instead of four per
// single rotation: at d = 20 a sweep is 57 phases, not 760.  Rotations of one round commute exactly (disjoint rows/columns).
static __device__ void lds_jacobi(double* A, double* V, double* cs, int d) {
    if (V) {
        for (int e = threadIdx.x; e < d * d; e += blockDim.x) V[e] = (e / d == e % d) ? 1.0 : 0.0;
    }
    wsync();
    const int np = (d + (d & 1)) - 1, half = (d + (d & 1)) / 2;
    int* pq = reinterpret_cast<int*>(cs + 32);       // 16 ints: (p << 8) | q of each slot of the round, -1 = idle
    double* xw = cs + 40;                            // 8 doubles: per-wave partial sums when the block has more than one wave
    const int kstep = blockDim.x >> 4, slstep = blockDim.x >> 5;
    for (int sweep = 0; sweep < 15; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
            const int r = e / d, c = e - r * d;
            const double x = A[e];
            if (r == c) dia = __builtin_fma(x, x, dia);
            else if (c < r) off = __builtin_fma(x, x, off);
        }
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_xor(off, o, 64);
            dia += __shfl_xor(dia, o, 64);
        }
        if (blockDim.x > 64) {            // up to 4 waves: combine the per-wave sums in a fixed order (block-uniform result)
            const int nw = blockDim.x >> 6;
            if ((threadIdx.x & 63) == 0) {
                xw[2 * (threadIdx.x >> 6)] = off;
                xw[2 * (threadIdx.x >> 6) + 1] = dia;
            }
            wsync();
            off = 0.0;
            dia = 0.0;
            for (int w = 0; w < nw; ++w) {
                off += xw[2 * w];
                dia += xw[2 * w + 1];
            }
            wsync();
        }
        if (off <= 1e-33 * dia) break;   // uniform: every thread holds the same sums
        for (int r = 0; r < np; ++r) {
            if ((int)threadIdx.x < half) {
                int p, q;
                jacobi_pair(r, threadIdx.x, np, p, q);
                double c = 1.0, sn = 0.0;
                if (q < d) {
                    const double apq = A[q * d + p], app = A[p * d + p], aqq = A[q * d + q];
                    const double h = aqq - app;
                    const double den = __builtin_fabs(h) + __builtin_sqrt(h * h + 4.0 * apq * apq);
                    double t = (den == 0.0) ? 0.0 : copysign_d(2.0 * apq, apq * h) / (den == 0.0 ? 1.0 : den);
                    if (h == 0.0) t = (apq == 0.0) ? 0.0 : copysign_d(1.0, apq);
                    c = 1.0 / __builtin_sqrt(t * t + 1.0);
                    sn = t * c;
                } else {
                    p = q = -1;          // the pair with the padding index sits out this round
                }
                cs[2 * threadIdx.x] = c;
                cs[2 * threadIdx.x + 1] = sn;
                pq[threadIdx.x] = (p << 8) | (q & 0xff);
            }
            wsync();
            // columns p, q of A and V for every pair of the round  (A <- A J): lane = (row group, pair), no integer division
            {
                const int sl = threadIdx.x & 15, k0 = threadIdx.x >> 4;
                const int code = sl < half ? pq[sl] : -1;
                if (code >= 0) {
                    const int p = code >> 8, q = code & 0xff;
                    const double c = cs[2 * sl], sn = cs[2 * sl + 1];
                    for (int k = k0; k < d; k += kstep) {
                        const double akp = A[k * d + p], akq = A[k * d + q];
                        A[k * d + p] = c * akp - sn * akq;
                        A[k * d + q] = sn * akp + c * akq;
                        if (V) {
                            const double vkp = V[k * d + p], vkq = V[k * d + q];
                            V[k * d + p] = c * vkp - sn * vkq;
                            V[k * d + q] = sn * vkp + c * vkq;
                        }
                    }
                }
            }
            wsync();
            // rows p, q of A  (A <- J^T A): lane = (pair group, column); the rotated pair itself is annihilated exactly
            {
                const int k = threadIdx.x & 31;
                if (k < d) {
                    for (int sl = threadIdx.x >> 5; sl < half; sl += slstep) {
                        const int code = pq[sl];
                        if (code < 0) continue;
                        const int p = code >> 8, q = code & 0xff;
                        const double c = cs[2 * sl], sn = cs[2 * sl + 1];
                        const double apk = A[p * d + k], aqk = A[q * d + k];
                        A[p * d + k] = (k == q) ? 0.0 : c * apk - sn * aqk;
                        A[q * d + k] = (k == p) ? 0.0 : sn * apk + c * aqk;
                    }
                }
            }
            wsync();
        }
    }
}

// Eigen-decomposition with the contract of lds_jacobi (eigenvalues on the diagonal of A, eigenvectors in the columns of V, V may be
// null), by the faster method for the order: Jacobi up to d = 8 (QL = false), Householder + QL in the registers of the block's first
// wave above (QL = true, kWaveEighMinDim <= d <= 32: wave_eigh.hpp).  Called by all threads of the block; ends with a barrier.
// QL is a template parameter of the calling KERNEL (chosen by its launcher from d): wave_eigh is a 289-register function, and a kernel
// that can reach it is allocated for it even when d is small.
template <bool QL>
static __device__ void lds_eigh(double* A, double* V, double* cs, int d) {
    if constexpr (QL) {
        if (threadIdx.x < 64) wave_eigh_any(A, V, cs, d);
        wsync();
    } else {
        lds_jacobi(A, V, cs, d);
    }
}

// F = V f(diag(A)) V^T.  fl: d doubles of LDS scratch for f(lambda_k) (the eigen-solver's scratch is free by now) - evaluating f inside
// the product loop costs d^3 transcendental calls instead of d (a 20 x 20 logm on one wave: 100 k of its 230 k cycles, round 3).
static __device__ void lds_fun_from_eig(const double* A, const double* V, double* F, int d, int fn, double* fl) {
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const double lam = A[k * d + k];
        fl[k] = fn == FN_LOG ? log(lam) : (fn == FN_EXP ? exp(lam) : __builtin_sqrt(lam));
    }
    wsync();
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        const double* vr = V + r * d;
        const double* vc = V + c * d;
        double s0 = 0.0, s1 = 0.0;                    // (two products in flight: see lds_mm)
        int k = 0;
        for (; k + 2 <= d; k += 2) {
            const double a0 = vr[k], a1 = vr[k + 1], f0 = fl[k], f1 = fl[k + 1], b0 = vc[k], b1 = vc[k + 1];
            s0 = __builtin_fma(a0 * f0, b0, s0);
            s1 = __builtin_fma(a1 * f1, b1, s1);
        }
        if (k < d) s0 = __builtin_fma(vr[k] * fl[k], vc[k], s0);
        F[e] = s0 + s1;
    }
    wsync();
}

static __device__ void lds_store(const double* A, double* __restrict__ dst, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) dst[e] = A[e];
}


}  // namespace gabo
