// Reconstruction cost of the nested-SPD mapping and its gradient in ONE launch (HD-GaBO, config 5):
//   cost(V, C, K) = sum_n dist(X_n, R [[Y_n, B_n], [B_n^T, C]] R^T)^2,   R = [W, V],  B_n = Y_n^1/2 K C^1/2,
// with the log-Euclidean distance ||logm X_n - logm Xrec_n + 1e-15||_F or the affine-invariant distance sqrt(sum log^2 lambda + 1e-15)
// (nested_mappings/nested_spd_optimization.py:23-92 on top of nested_spd_utils.py:51-118, spd_utils_torch.py:13-50, 53-156).
// The reference evaluates it with a Python loop over the data and differentiates it by autograd; the augmented-Lagrangian optimiser
// (nested_spd_optimization.py:95-186) asks for ~600 values / gradients per BO iteration, and round 2 served each with ~40 small
// launches (two eigen-solves, a dozen products and their adjoints replayed from a hipGraph: 1.07 ms per evaluation at D = 20).
//
// Grid: one block of 256 threads per (parameter set p, data point n).  Every block factors C itself (sqrtm through wave_eigh: the
// factorisation is on the critical path of every block anyway, so repeating it costs no time and saves a launch + a grid-wide wait),
// builds its reconstruction, takes its eigen-decomposition, and pushes the adjoint back to per-block partial gradients in the
// workspace.  The LAST block of a parameter set to finish (an atomic ticket) adds the partials in data order - the result does not
// depend on the order in which the blocks ran - and applies the adjoint of C -> C^1/2 with the factorisation it still holds in LDS.
// Everything is d^3-sized and latency-bound: two dependent eigen-solves of order D - d and D (~50 us each at D = 20, one wave) and ~10
// products spread over the 256 threads.
#include "gabo_device.hpp"
#include "lds_linalg.hpp"
#include "../../include/gabo_hip.h"

// Development instrumentation (-DGABO_RECON_CLOCKS, tools/recon_clocks.py): block 0 / thread 0 stores the shader clock at the phase boundaries.
#ifdef GABO_RECON_CLOCKS
static __device__ long long gabo_recon_clk[16];
#define GABO_RECON_TICK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) gabo_recon_clk[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GABO_RECON_TICK(i) do { } while (0)
#endif

namespace gabo {

// C[M x N] = op(A) op(B) with element strides: A(i, k) = A[i * ai + k * ak], B(k, j) = B[k * bk + j * bj]; C row-major (ldc = N);
// scale applied to the result.  All threads of the block; ends with a barrier.
static __device__ void lds_gemm(const double* A, int ai, int ak, const double* B, int bk, int bj, double* C, int M, int N, int K,
                                double scale = 1.0) {
    for (int e = threadIdx.x; e < M * N; e += blockDim.x) {
        const int i = e / N, j = e - i * N;
        const double* ap = A + i * ai;
        const double* bp = B + j * bj;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;          // (four products in flight: see lds_mm)
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            const double a0 = ap[k * ak], a1 = ap[(k + 1) * ak], a2 = ap[(k + 2) * ak], a3 = ap[(k + 3) * ak];
            const double b0 = bp[k * bk], b1 = bp[(k + 1) * bk], b2 = bp[(k + 2) * bk], b3 = bp[(k + 3) * bk];
            s0 = __builtin_fma(a0, b0, s0);
            s1 = __builtin_fma(a1, b1, s1);
            s2 = __builtin_fma(a2, b2, s2);
            s3 = __builtin_fma(a3, b3, s3);
        }
        for (; k < K; ++k) s0 = __builtin_fma(ap[k * ak], bp[k * bk], s0);
        C[e] = ((s0 + s1) + (s2 + s3)) * scale;
    }
    __syncthreads();
}

// divided differences of log / sqrt at two eigenvalues (the diagonal gives the derivative) from the eigenvalues and f at them; the same
// expressions as spd_matfun_backward_kernel (spd_manifold.hip)
static __device__ __forceinline__ double divided_difference(double lr, double lc, double fr, double fc, int fn) {
    if (fn == FN_LOG) {
        const double mean = 0.5 * (lr + lc), dl = lr - lc;
        const double z = dl / (2.0 * mean), z2 = z * z;
        return (__builtin_fabs(z) < 1e-3) ? (1.0 + z2 * (1.0 / 3.0 + z2 * (0.2 + z2 * (1.0 / 7.0)))) / mean : (fr - fc) / dl;
    }
    return 1.0 / (fr + fc);
}

// out = U ((U^T sym(G) U) o F) U^T, symmetrised; lam on the diagonal of Lam.  G is overwritten; T scratch; fl: n doubles of scratch for f(lambda).
// n x n matrices in LDS.
static __device__ void lds_matfun_adjoint(const double* Lam, const double* U, double* G, double* T, double* out, int n, int fn, double* fl) {
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double lam = Lam[k * n + k];
        fl[k] = fn == FN_LOG ? log(lam) : __builtin_sqrt(lam);
    }
    lds_symmetrize(G, T, n);
    lds_mm(U, G, T, n, true, false);          // U^T G
    lds_mm(T, U, G, n, false, false);         // U^T G U
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int r = e / n, c = e - r * n;
        G[e] *= divided_difference(Lam[r * n + r], Lam[c * n + c], fl[r], fl[c], fn);
    }
    __syncthreads();
    lds_mm(U, G, T, n, false, false);
    lds_mm(T, U, out, n, false, true);
    lds_symmetrize(out, T, n);
}

struct ReconLayout {
    int D, d, m;
    // per-block partial record in the workspace: [cost, GV (D m), Gbb (m m), GT (d m)]
    __host__ __device__ int record() const { return 1 + D * m + m * m + d * m; }
    __host__ __device__ size_t lds_doubles() const {
        return (size_t)5 * D * D + (size_t)3 * D * m + (size_t)4 * m * m + (size_t)D * d + (size_t)4 * d * m + (size_t)2 * d * d + kJacobiScratch + 8;
    }
};

// data: N x D x D, what the metric needs of X_n: logm X_n (metric 1) or chol(X_n)^-1 (metric 0); y, sy: N x d x d (Y_n and its square root);
// w: D x d; v, c, k: P parameter sets; cost: P; gv, gc, gk: P gradients or all null; ws: counters (P ints, zeroed) then P N records.
__global__ __launch_bounds__(256) void nested_spd_reconstruction_kernel(const double* __restrict__ data, const double* __restrict__ y,
                                                                        const double* __restrict__ sy, const double* __restrict__ w,
                                                                        const double* __restrict__ v, const double* __restrict__ c,
                                                                        const double* __restrict__ k, double* __restrict__ cost,
                                                                        double* __restrict__ gv, double* __restrict__ gc,
                                                                        double* __restrict__ gk, const double* __restrict__ c_lam,
                                                                        const double* __restrict__ c_vec, int P, int N, int D, int d,
                                                                        int metric, int* __restrict__ counters, double* __restrict__ records) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = D - d, DD = D * D, mm = m * m;
    const ReconLayout lay{D, d, m};
    const int rec = lay.record();
    double* M0 = lds;                 // D x D work matrices
    double* M1 = M0 + DD;
    double* M2 = M1 + DD;
    double* M3 = M2 + DD;
    double* M4 = M3 + DD;
    double* Vl = M4 + DD;             // D x m
    double* P1 = Vl + D * m;          // D x m : W B + V C
    double* Q1 = P1 + D * m;          // D x m : W B, later G V
    double* Cl = Q1 + D * m;          // m x m : sym(C)
    double* Lc = Cl + mm;             // m x m : eigenvalues of C on the diagonal
    double* Uc = Lc + mm;             // m x m : eigenvectors of C
    double* Cs = Uc + mm;             // m x m : C^1/2
    double* Wl = Cs + mm;             // D x d
    double* Kl = Wl + D * d;          // d x m
    double* Tl = Kl + d * m;          // d x m : K C^1/2
    double* Bl = Tl + d * m;          // d x m : Y^1/2 K C^1/2
    double* GB = Bl + d * m;          // d x m
    double* Yl = GB + d * m;          // d x d
    double* Sl = Yl + d * d;          // d x d
    double* cs = Sl + d * d;          // eigen-solver scratch
    double* red = cs + kJacobiScratch;  // 8 doubles: block reductions
    const int p = blockIdx.x / N, n = blockIdx.x - p * N;
    const bool want_grad = gv != nullptr;

    // ---- inputs
    GABO_RECON_TICK(0);
    for (int e = threadIdx.x; e < D * d; e += blockDim.x) Wl[e] = w[e];
    for (int e = threadIdx.x; e < D * m; e += blockDim.x) Vl[e] = v[(size_t)p * D * m + e];
    for (int e = threadIdx.x; e < d * m; e += blockDim.x) Kl[e] = k[(size_t)p * d * m + e];
    for (int e = threadIdx.x; e < mm; e += blockDim.x) {
        const int r = e / m, cc = e - r * m;
        const double* cp = c + (size_t)p * mm;
        Cl[e] = 0.5 * (cp[r * m + cc] + cp[cc * m + r]);
    }
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        const int r = e / d, cc = e - r * d;
        Yl[e] = 0.5 * (y[(size_t)n * d * d + r * d + cc] + y[(size_t)n * d * d + cc * d + r]);
        Sl[e] = 0.5 * (sy[(size_t)n * d * d + r * d + cc] + sy[(size_t)n * d * d + cc * d + r]);
    }
    __syncthreads();
    // ---- C^1/2 = Uc sqrt(Lc) Uc^T: from the caller's eigen-decomposition of C when it has one (the native optimisation loop factors
    // each iterate on the host in ~5 us; here it is a lone wave's ~50 us on the critical path of every block), else factored here
    GABO_RECON_TICK(1);
    if (c_lam != nullptr) {
        for (int e = threadIdx.x; e < mm; e += blockDim.x) {
            const int r = e / m, cc = e - r * m;
            Lc[e] = r == cc ? c_lam[(size_t)p * m + r] : 0.0;
            Uc[e] = c_vec[(size_t)p * mm + e];
        }
        __syncthreads();
    } else {
        for (int e = threadIdx.x; e < mm; e += blockDim.x) Lc[e] = Cl[e];
        __syncthreads();
        if (m >= kWaveEighMinDim) lds_eigh<true>(Lc, Uc, cs, m);
        else lds_eigh<false>(Lc, Uc, cs, m);
    }
    GABO_RECON_TICK(2);
    lds_fun_from_eig(Lc, Uc, Cs, m, FN_SQRT, cs);
    lds_gemm(Kl, m, 1, Cs, m, 1, Tl, d, m, m);            // T = K C^1/2
    lds_gemm(Sl, d, 1, Tl, m, 1, Bl, d, m, d);            // B = Y^1/2 T
    lds_gemm(Wl, d, 1, Bl, m, 1, Q1, D, m, d);            // Q1 = W B
    for (int e = threadIdx.x; e < D * m; e += blockDim.x) {
        const int r = e / m, j = e - r * m;
        double s = Q1[e];
        for (int q = 0; q < m; ++q) s = __builtin_fma(Vl[r * m + q], Cl[q * m + j], s);
        P1[e] = s;                                          // W B + V C
    }
    lds_gemm(Wl, d, 1, Yl, d, 1, M1, D, d, d);            // M1 (D x d) = W Y
    // ---- Xrec = W Y W^T + P1 V^T + V Q1^T   (lower triangle, mirrored: exactly symmetric)
    for (int e = threadIdx.x; e < DD; e += blockDim.x) {
        const int r = e / D, cc = e - r * D;
        if (cc > r) continue;
        double s = 0.0;
        for (int a = 0; a < d; ++a) s = __builtin_fma(M1[r * d + a], Wl[cc * d + a], s);
        for (int j = 0; j < m; ++j) s = __builtin_fma(P1[r * m + j], Vl[cc * m + j], __builtin_fma(Vl[r * m + j], Q1[cc * m + j], s));
        M0[r * D + cc] = s;
        M0[cc * D + r] = s;
    }
    __syncthreads();
    // ---- the matrix whose spectrum is the distance: Xrec (log-Euclidean) or L_n^-1 Xrec L_n^-T (affine-invariant)
    if (metric == 0) {
        lds_load(data + (size_t)n * DD, M4, D);           // L_n^-1
        lds_congruence(M4, M0, M2, M1, D);
        lds_symmetrize(M2, M1, D);
        for (int e = threadIdx.x; e < DD; e += blockDim.x) M0[e] = M2[e];
        __syncthreads();
    }
    GABO_RECON_TICK(3);
    if (D >= kWaveEighMinDim) lds_eigh<true>(M0, M1, cs, D);      // M0: eigenvalues on the diagonal, M1 = U
    else lds_eigh<false>(M0, M1, cs, D);
    GABO_RECON_TICK(4);
    double part = 0.0;
    if (metric == 0) {
        // cost_n = sum log^2 lambda + 1e-15 ;  G_M = U diag(2 log lambda / lambda) U^T ;  G = L^-T G_M L^-1
        if (threadIdx.x == 0) {
            double s = 1e-15;
            for (int q = 0; q < D; ++q) { const double lg = log(M0[q * D + q]); s = __builtin_fma(lg, lg, s); }
            part = s;
        }
        if (want_grad) {
            for (int q = threadIdx.x; q < D; q += blockDim.x) {
                const double lam = M0[q * D + q];
                cs[q] = 2.0 * log(lam) / lam;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < DD; e += blockDim.x) {
                const int r = e / D, cc = e - r * D;
                double s = 0.0;
                for (int q = 0; q < D; ++q) s = __builtin_fma(M1[r * D + q] * cs[q], M1[cc * D + q], s);
                M2[e] = s;
            }
            __syncthreads();
            lds_mm(M4, M2, M3, D, true, false);            // L^-T G_M
            lds_mm(M3, M4, M2, D, false, false);           // L^-T G_M L^-1
            lds_symmetrize(M2, M3, D);                     // M2 = G
        }
    } else {
        lds_fun_from_eig(M0, M1, M2, D, FN_LOG, cs);          // logm Xrec
        for (int e = threadIdx.x; e < DD; e += blockDim.x) {
            const double diff = (data[(size_t)n * DD + e] - M2[e]) + 1e-15;
            part = __builtin_fma(diff, diff, part);
            M2[e] = -2.0 * diff;                           // d cost / d logm Xrec
        }
        __syncthreads();
        if (want_grad) {
            lds_matfun_adjoint(M0, M1, M2, M3, M4, D, FN_LOG, cs);
            for (int e = threadIdx.x; e < DD; e += blockDim.x) M2[e] = M4[e];   // M2 = G
            __syncthreads();
        }
    }
    GABO_RECON_TICK(5);
    // block sum of the cost in a fixed order
    {
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
    }
    double* R = records + ((size_t)p * N + n) * rec;
    if (threadIdx.x == 0) R[0] = (red[0] + red[1]) + (red[2] + red[3]);
    if (want_grad) {
        // G_V = 2 G P1 ;  Q1 <- G V ;  Gbb = V^T (G V) ;  G_B = 2 W^T (G V) ;  G_T = Y^1/2 G_B
        double* RV = R + 1;
        double* Rbb = RV + D * m;
        double* RT = Rbb + mm;
        for (int e = threadIdx.x; e < D * m; e += blockDim.x) {
            const int r = e / m, j = e - r * m;
            double s1 = 0.0, s2 = 0.0;
            for (int q = 0; q < D; ++q) {
                const double g = M2[r * D + q];
                s1 = __builtin_fma(g, P1[q * m + j], s1);
                s2 = __builtin_fma(g, Vl[q * m + j], s2);
            }
            RV[e] = 2.0 * s1;
            Q1[e] = s2;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < mm; e += blockDim.x) {
            const int r = e / m, j = e - r * m;
            double s = 0.0;
            for (int q = 0; q < D; ++q) s = __builtin_fma(Vl[q * m + r], Q1[q * m + j], s);
            Rbb[e] = s;
        }
        lds_gemm(Wl, 1, d, Q1, m, 1, GB, d, m, D, 2.0);   // G_B = 2 W^T (G V)
        for (int e = threadIdx.x; e < d * m; e += blockDim.x) {
            const int r = e / m, j = e - r * m;
            double s = 0.0;
            for (int q = 0; q < d; ++q) s = __builtin_fma(Sl[r * d + q], GB[q * m + j], s);
            RT[e] = s;
        }
    }
    GABO_RECON_TICK(6);
    // ---- the last block of this parameter set adds the partials in data order and finishes the chain through C^1/2
    __threadfence();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = (atomicAdd(counters + p, 1) == N - 1) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    __threadfence();
    const double* Rp = records + (size_t)p * N * rec;
    if (threadIdx.x == 0) {
        counters[p] = 0;                  // every block of this parameter set has drawn its ticket: the next launch finds it at zero
        double s = 0.0;
        for (int q = 0; q < N; ++q) s += Rp[(size_t)q * rec];
        cost[p] = s;
    }
    if (!want_grad) return;
    double* SumBB = M0;               // m x m
    double* SumT = M1;                // d x m
    for (int e = threadIdx.x; e < D * m + mm + d * m; e += blockDim.x) {
        // (four records in flight: one dependent global load per record was ~2 k cycles each on the tail of every gradient launch; the order
        // of the additions is fixed - it does not depend on the order in which the blocks finished)
        const double* rp = Rp + 1 + e;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int q = 0;
        for (; q + 4 <= N; q += 4) {
            const double v0 = rp[(size_t)q * rec], v1 = rp[(size_t)(q + 1) * rec], v2 = rp[(size_t)(q + 2) * rec], v3 = rp[(size_t)(q + 3) * rec];
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        for (; q < N; ++q) s0 += rp[(size_t)q * rec];
        const double s = (s0 + s1) + (s2 + s3);
        if (e < D * m) gv[(size_t)p * D * m + e] = s;
        else if (e < D * m + mm) SumBB[e - D * m] = s;
        else SumT[e - D * m - mm] = s;
    }
    __syncthreads();
    // T = K C^1/2:  G_K = G_T C^1/2 ,  G_Cs = K^T G_T ;  C^1/2 = sqrtm(C):  G_C = Gbb + adjoint(G_Cs)
    for (int e = threadIdx.x; e < d * m; e += blockDim.x) {
        const int r = e / m, j = e - r * m;
        double s = 0.0;
        for (int q = 0; q < m; ++q) s = __builtin_fma(SumT[r * m + q], Cs[q * m + j], s);
        gk[(size_t)p * d * m + e] = s;
    }
    lds_gemm(Kl, 1, m, SumT, m, 1, M2, m, m, d);          // K^T G_T
    lds_matfun_adjoint(Lc, Uc, M2, M3, M4, m, FN_SQRT, cs);
    for (int e = threadIdx.x; e < mm; e += blockDim.x) gc[(size_t)p * mm + e] = SumBB[e] + M4[e];
    GABO_RECON_TICK(7);
}

// what the cost needs of the data, once per data set: logm X_n (metric 1) or chol(X_n)^-1 (metric 0)
template <bool QL>
__global__ __launch_bounds__(64) void nested_spd_reconstruction_prepare_kernel(const double* __restrict__ x, double* __restrict__ out, int D,
                                                                               int metric, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int DD = D * D;
    double* M0 = lds;
    double* M1 = M0 + DD;
    double* M2 = M1 + DD;
    double* cs = M2 + DD;
    const size_t n = blockIdx.x;
    lds_load(x + n * DD, M0, D);
    lds_symmetrize(M0, M1, D);
    if (metric == 0) {
        const bool ok = lds_cholesky(M0, D);
        lds_tri_inverse(M0, M1, D);
        lds_store(M1, out + n * DD, D);
        if (!ok && threadIdx.x == 0 && status) {
            if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)n;
        }
    } else {
        lds_eigh<QL>(M0, M1, cs, D);
        lds_fun_from_eig(M0, M1, M2, D, FN_LOG, cs);
        lds_store(M2, out + n * DD, D);
    }
}

int nested_spd_reconstruction_launch(const double* data, const double* y, const double* sqrt_y, const double* w, const double* v,
                                     const double* c, const double* k, double* cost, double* grad_v, double* grad_c, double* grad_k,
                                     const double* c_eigenvalues, const double* c_eigenvectors, int64_t P, int64_t N, int D, int d, int metric,
                                     void* workspace, size_t workspace_bytes, bool clear_tickets, gabo_stream_t stream);

}  // namespace gabo

extern "C" {

#ifdef GABO_RECON_CLOCKS
int gabo_debug_recon_clocks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gabo_recon_clk), 16 * sizeof(long long)); }
#endif

size_t gabo_nested_spd_reconstruction_workspace_bytes(int64_t P, int64_t N, int D, int d) {
    if (P <= 0 || N <= 0 || D < 2 || d < 1 || d >= D) return 0;
    const gabo::ReconLayout lay{D, d, D - d};
    const size_t counters = ((size_t)P * sizeof(int) + 15) / 16 * 16;
    return counters + (size_t)P * (size_t)N * (size_t)lay.record() * sizeof(double);
}

int gabo_nested_spd_reconstruction_prepare(const double* x, double* out, int64_t N, int D, int metric, int* status, gabo_stream_t stream) {
    if (D < 2 || D > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (N < 0 || (metric != 0 && metric != 1)) return GABO_ERR_ARG;
    if (N == 0) return GABO_OK;
    if (!x || !out || N > 0x7fffffffLL) return GABO_ERR_ARG;
    const size_t lds = (size_t)(3 * D * D + gabo::kJacobiScratch) * sizeof(double);
    if (D >= gabo::kWaveEighMinDim)
        hipLaunchKernelGGL(gabo::nested_spd_reconstruction_prepare_kernel<true>, dim3((unsigned)N), dim3(64), lds, (hipStream_t)stream, x, out, D, metric, status);
    else
        hipLaunchKernelGGL(gabo::nested_spd_reconstruction_prepare_kernel<false>, dim3((unsigned)N), dim3(64), lds, (hipStream_t)stream, x, out, D, metric, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_nested_spd_reconstruction(const double* data, const double* y, const double* sqrt_y, const double* w, const double* v,
                                   const double* c, const double* k, double* cost, double* grad_v, double* grad_c, double* grad_k,
                                   const double* c_eigenvalues, const double* c_eigenvectors, int64_t P, int64_t N, int D, int d, int metric,
                                   void* workspace, size_t workspace_bytes, gabo_stream_t stream) {
    return gabo::nested_spd_reconstruction_launch(data, y, sqrt_y, w, v, c, k, cost, grad_v, grad_c, grad_k, c_eigenvalues, c_eigenvectors, P, N, D,
                                                  d, metric, workspace, workspace_bytes, true, stream);
}
}

namespace gabo {

// The launch behind gabo_nested_spd_reconstruction.  clear_tickets = false: the caller guarantees that the tickets at the head of the
// workspace are zero (they are after every completed launch: the last block of a parameter set puts its ticket back) - the native
// optimisation loop clears them once and saves a memset node per evaluation.
int nested_spd_reconstruction_launch(const double* data, const double* y, const double* sqrt_y, const double* w, const double* v,
                                     const double* c, const double* k, double* cost, double* grad_v, double* grad_c, double* grad_k,
                                     const double* c_eigenvalues, const double* c_eigenvectors, int64_t P, int64_t N, int D, int d, int metric,
                                     void* workspace, size_t workspace_bytes, bool clear_tickets, gabo_stream_t stream) {
    if (D < 2 || D > GABO_SPD_MAX_DIM || d < 1 || d >= D) return GABO_ERR_DIM;
    if (P < 0 || N < 0 || (metric != 0 && metric != 1) || ((c_eigenvalues == nullptr) != (c_eigenvectors == nullptr))) return GABO_ERR_ARG;
    if (P == 0) return GABO_OK;
    if (!cost) return GABO_ERR_ARG;
    const bool grads = grad_v || grad_c || grad_k;
    if (grads && !(grad_v && grad_c && grad_k)) return GABO_ERR_ARG;
    if (N == 0) {
        hipMemsetAsync(cost, 0, (size_t)P * sizeof(double), (hipStream_t)stream);
        if (grads) {
            hipMemsetAsync(grad_v, 0, (size_t)P * D * (D - d) * sizeof(double), (hipStream_t)stream);
            hipMemsetAsync(grad_c, 0, (size_t)P * (D - d) * (D - d) * sizeof(double), (hipStream_t)stream);
            hipMemsetAsync(grad_k, 0, (size_t)P * d * (D - d) * sizeof(double), (hipStream_t)stream);
        }
        return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
    }
    if (!data || !y || !sqrt_y || !w || !v || !c || !k || !workspace) return GABO_ERR_ARG;
    if (P * N > 0x7fffffffLL || workspace_bytes < gabo_nested_spd_reconstruction_workspace_bytes(P, N, D, d)) return GABO_ERR_ARG;
    const ReconLayout lay{D, d, D - d};
    const size_t counters = ((size_t)P * sizeof(int) + 15) / 16 * 16;
    int* cnt = static_cast<int*>(workspace);
    double* records = reinterpret_cast<double*>(static_cast<char*>(workspace) + counters);
    if (clear_tickets) hipMemsetAsync(cnt, 0, counters, (hipStream_t)stream);
    const size_t lds = lay.lds_doubles() * sizeof(double);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(nested_spd_reconstruction_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return GABO_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(nested_spd_reconstruction_kernel, dim3((unsigned)(P * N)), dim3(256), lds, (hipStream_t)stream, data, y, sqrt_y, w, v,
                       c, k, cost, grad_v, grad_c, grad_k, c_eigenvalues, c_eigenvectors, (int)P, (int)N, D, d, metric, cnt, records);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // namespace gabo
