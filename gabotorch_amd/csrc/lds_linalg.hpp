// Small dense linear algebra on d x d tiles held in LDS, executed cooperatively by the threads of one block (one wave):
// the building blocks of the wave-per-matrix kernels (spd_manifold.hip) and of the wave-per-pair fallback for d > 12
// (spd_pairwise_generic.hip).  Every function is called by all threads of the block and ends with a barrier.
#pragma once
#include "gabo_device.hpp"

namespace gabo {

enum { FN_LOG = 0, FN_EXP = 1, FN_SQRT = 2 };

static __device__ __forceinline__ void wsync() { __syncthreads(); }

// symmetric matrix from a Mandel vector (global) into LDS
static __device__ void lds_from_mandel(const double* __restrict__ v, double* A, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        int hi = r > c ? r : c, lo = r > c ? c : r;
        double x = v[mandel_pos(d, hi, lo)];
        A[e] = (r == c) ? x : x / kSqrt2;
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
        double lcc = __builtin_sqrt(piv);
        wsync();
        for (int r = c + threadIdx.x; r < d; r += blockDim.x) {
            if (r == c) {
                A[c * d + c] = lcc;
            } else {
                double s = A[r * d + c];
                for (int k = 0; k < c; ++k) s -= A[r * d + k] * A[c * d + k];
                A[r * d + c] = s / lcc;
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
        W[c * d + c] = 1.0 / L[c * d + c];
        for (int r = c + 1; r < d; ++r) {
            double s = 0.0;
            for (int k = c; k < r; ++k) s += L[r * d + k] * W[k * d + c];
            W[r * d + c] = -s / L[r * d + r];
        }
    }
    wsync();
}

// C = op(A) op(B), thread per output element; C must not alias A or B
static __device__ void lds_mm(const double* A, const double* B, double* C, int d, bool ta, bool tb) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) {
            double a = ta ? A[k * d + r] : A[r * d + k];
            double b = tb ? B[c * d + k] : B[k * d + c];
            s = __builtin_fma(a, b, s);
        }
        C[e] = s;
    }
    wsync();
}

// C = A B A^T (congruence), T scratch
static __device__ void lds_congruence(const double* A, const double* B, double* C, double* T, int d) {
    lds_mm(A, B, T, d, false, false);
    lds_mm(T, A, C, d, false, true);
}

// Cyclic Jacobi: A (symmetric, full storage) -> diagonal; V = eigenvectors in columns (V may be null: eigenvalues only).
// cs: 2 doubles of LDS.
static __device__ void lds_jacobi(double* A, double* V, double* cs, int d) {
    if (V) {
        for (int e = threadIdx.x; e < d * d; e += blockDim.x) V[e] = (e / d == e % d) ? 1.0 : 0.0;
    }
    wsync();
    for (int sweep = 0; sweep < 15; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int r = 0; r < d; ++r) {
            dia += A[r * d + r] * A[r * d + r];
            for (int c = 0; c < r; ++c) off += A[r * d + c] * A[r * d + c];
        }
        if (off <= 1e-33 * dia) break;   // uniform: every thread reads the same LDS values
        for (int p = 0; p < d - 1; ++p) {
            for (int q = p + 1; q < d; ++q) {
                if (threadIdx.x == 0) {
                    double apq = A[q * d + p], app = A[p * d + p], aqq = A[q * d + q];
                    double h = aqq - app;
                    double den = __builtin_fabs(h) + __builtin_sqrt(h * h + 4.0 * apq * apq);
                    double t = (den == 0.0) ? 0.0 : copysign_d(2.0 * apq, apq * h) / (den == 0.0 ? 1.0 : den);
                    if (h == 0.0) t = (apq == 0.0) ? 0.0 : copysign_d(1.0, apq);
                    double c = 1.0 / __builtin_sqrt(t * t + 1.0);
                    cs[0] = c;
                    cs[1] = t * c;
                }
                wsync();
                double c = cs[0], s = cs[1];
                // columns p, q of A and V  (A <- A J)
                for (int k = threadIdx.x; k < d; k += blockDim.x) {
                    double akp = A[k * d + p], akq = A[k * d + q];
                    A[k * d + p] = c * akp - s * akq;
                    A[k * d + q] = s * akp + c * akq;
                    if (V) {
                        double vkp = V[k * d + p], vkq = V[k * d + q];
                        V[k * d + p] = c * vkp - s * vkq;
                        V[k * d + q] = s * vkp + c * vkq;
                    }
                }
                wsync();
                // rows p, q of A  (A <- J^T A)
                for (int k = threadIdx.x; k < d; k += blockDim.x) {
                    double apk = A[p * d + k], aqk = A[q * d + k];
                    A[p * d + k] = c * apk - s * aqk;
                    A[q * d + k] = s * apk + c * aqk;
                }
                wsync();
                if (threadIdx.x == 0) { A[p * d + q] = 0.0; A[q * d + p] = 0.0; }
                wsync();
            }
        }
    }
}


// F = V f(diag(A)) V^T
static __device__ void lds_fun_from_eig(const double* A, const double* V, double* F, int d, int fn) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) {
            double lam = A[k * d + k];
            double f = fn == FN_LOG ? log(lam) : (fn == FN_EXP ? exp(lam) : __builtin_sqrt(lam));
            s = __builtin_fma(V[r * d + k] * f, V[c * d + k], s);
        }
        F[e] = s;
    }
    wsync();
}

static __device__ void lds_store(const double* A, double* __restrict__ dst, int d) {
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) dst[e] = A[e];
}


}  // namespace gabo
