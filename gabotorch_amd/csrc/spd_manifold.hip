// Batched Riemannian operations on the SPD manifold for the acquisition maximiser: one WAVE per matrix, d x d tiles
// staged in LDS, runtime d (2..32).  These are the per-restart operations the reference calls through pymanopt's
// PositiveDefinite manifold and its own numpy maps:
//   exp / retr   X expm(X^-1 U)                       Riemannian_utils/spd_utils.py:104-120 ; [3P] PositiveDefinite.exp/retr
//   log          L logm(L^-1 Y L^-T) L^T              spd_utils.py:123-139 ; pymanopt_addons/tools/multi.py:55-64
//   inner/norm   tr(X^-1 U X^-1 V)                    [3P] PositiveDefinite.inner/norm  (call sites robust_trust_regions.py:148,175)
//   dist         ||logm(L^-1 Y L^-T)||_F              [3P] PositiveDefinite.dist (gabo_spd.py:289)
//   egrad2rgrad  X sym(G) X                           [3P] (pymanopt_addons/problem.py:135)
//   ehess2rhess  X sym(H) X + sym(U sym(G) X)         [3P] (problem.py:156)
//   logm / expm / sqrtm of a symmetric matrix         spd_utils_torch.py:13-50 ; tools/multi.py:55-75
//   lambda_max / lambda_min and their gradients       spd_constraints_utils_torch.py:17-50
// R restarts are R independent blocks; inside a block the lanes share the O(d^3) loops element-wise; the eigen-decomposition is the
// parallel-ordering Jacobi of lds_linalg.hpp up to d = 4 and Householder + QL with one matrix row per lane above (wave_eigh.hpp:
// 118 k against 634 k shader cycles at d = 20, tools/ubench_eigh.hip).
#include "gabo_device.hpp"
#include "lds_linalg.hpp"
#include "spd_eigvec.hpp"
#include "spd_project_operator.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

// op codes of gabo_spd_manifold_op
enum {
    OP_EXP = 0, OP_LOG = 1, OP_INNER = 2, OP_NORM = 3, OP_DIST = 4, OP_EGRAD2RGRAD = 5, OP_EHESS2RHESS = 6, OP_LOGM = 7,
    OP_EXPM = 8, OP_SQRTM = 9, OP_EIGMAX = 10, OP_EIGMIN = 11
};

// a, b, c: up to three input matrix batches (n x d x d, row-major); out: n x d x d or n scalars (+ n x d x d gradient in out2)
// THREADS = 64: one wave per matrix (the compiler drops the barriers of a single-wave workgroup); 256: four waves for d > 12 and a
// small batch.  QL: the eigen-solver this instantiation links (lds_eigh<QL>).
template <int THREADS, bool QL>
__global__ __launch_bounds__(THREADS) void spd_manifold_kernel(int op, const double* __restrict__ a, const double* __restrict__ b,
                                                          const double* __restrict__ c, const double* __restrict__ e,
                                                          double* __restrict__ out, double* __restrict__ out2, int64_t n, int d,
                                                          int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* M0 = lds;            // 6 matrices + the Jacobi scratch
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* M4 = M3 + dd;
    double* M5 = M4 + dd;
    double* cs = M5 + dd;
    const int64_t i = blockIdx.x;
    const double* A = a + i * dd;
    bool ok = true;
    switch (op) {
        case OP_EXP:    // a = X (base), b = U (tangent)
        case OP_LOG: {  // a = X (base), b = Y (point)
            lds_load(A, M0, d);
            lds_load(b + i * dd, M1, d);
            lds_symmetrize(M1, M5, d);
            ok = lds_cholesky(M0, d);            // M0 = L
            lds_tri_inverse(M0, M2, d);          // M2 = W
            lds_congruence(M2, M1, M3, M4, d);   // M3 = W B W^T
            lds_symmetrize(M3, M5, d);
            lds_eigh<QL>(M3, M4, cs, d);           // M3 diag, M4 = V
            lds_fun_from_eig(M3, M4, M1, d, op == OP_EXP ? FN_EXP : FN_LOG, cs);
            lds_congruence(M0, M1, M3, M4, d);   // L F L^T
            lds_symmetrize(M3, M5, d);
            lds_store(M3, out + i * dd, d);
            break;
        }
        case OP_INNER:   // a = X, b = U, c = V
        case OP_NORM: {  // a = X, b = U
            lds_load(A, M0, d);
            lds_load(b + i * dd, M1, d);
            ok = lds_cholesky(M0, d);
            lds_tri_inverse(M0, M2, d);
            lds_congruence(M2, M1, M3, M4, d);
            if (op == OP_INNER) {
                lds_load(c + i * dd, M1, d);
                lds_congruence(M2, M1, M5, M4, d);
            }
            const double* Q = op == OP_INNER ? M5 : M3;
            if (threadIdx.x == 0) {
                double s = 0.0;
                for (int k = 0; k < dd; ++k) s = __builtin_fma(M3[k], Q[k], s);
                out[i] = op == OP_INNER ? s : __builtin_sqrt(s > 0.0 ? s : 0.0);
            }
            break;
        }
        case OP_DIST: {  // a = X, b = Y
            lds_load(A, M0, d);
            lds_load(b + i * dd, M1, d);
            ok = lds_cholesky(M0, d);
            lds_tri_inverse(M0, M2, d);
            lds_congruence(M2, M1, M3, M4, d);
            lds_symmetrize(M3, M5, d);
            lds_eigh<QL>(M3, M4, cs, d);
            if (threadIdx.x == 0) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) { double lg = log(M3[k * d + k]); s = __builtin_fma(lg, lg, s); }
                out[i] = __builtin_sqrt(s);
            }
            break;
        }
        case OP_EGRAD2RGRAD: {  // a = X, b = G  ->  X sym(G) X
            lds_load(A, M0, d);
            lds_load(b + i * dd, M1, d);
            lds_symmetrize(M1, M5, d);
            lds_congruence(M0, M1, M3, M4, d);   // X symmetric: X S X^T = X S X
            lds_store(M3, out + i * dd, d);
            break;
        }
        case OP_EHESS2RHESS: {  // a = X, b = egrad, c = ehess, e = U  ->  X sym(eh) X + sym(U sym(eg) X)
            lds_load(A, M0, d);
            lds_load(c + i * dd, M1, d);
            lds_symmetrize(M1, M5, d);
            lds_congruence(M0, M1, M3, M4, d);   // M3 = X sym(eh) X
            lds_load(b + i * dd, M1, d);
            lds_symmetrize(M1, M5, d);           // M1 = sym(eg)
            lds_load(e + i * dd, M2, d);         // M2 = U
            lds_mm(M2, M1, M4, d, false, false); // U sym(eg)
            lds_mm(M4, M0, M5, d, false, false); // U sym(eg) X
            for (int k = threadIdx.x; k < dd; k += blockDim.x) {
                int r = k / d, cc = k - r * d;
                out[i * dd + k] = M3[k] + 0.5 * (M5[k] + M5[cc * d + r]);
            }
            break;
        }
        case OP_LOGM:
        case OP_EXPM:
        case OP_SQRTM: {
            lds_load(A, M0, d);
            lds_symmetrize(M0, M5, d);
            lds_eigh<QL>(M0, M1, cs, d);
            lds_fun_from_eig(M0, M1, M2, d, op == OP_LOGM ? FN_LOG : (op == OP_EXPM ? FN_EXP : FN_SQRT), cs);
            lds_store(M2, out + i * dd, d);
            if (out2) {          // hand the eigen-decomposition to gabo_spd_matfun_backward_eig: V (d x d), then the d eigenvalues
                double* eg = out2 + i * (dd + d);
                for (int k = threadIdx.x; k < dd; k += blockDim.x) eg[k] = M1[k];
                for (int k = threadIdx.x; k < d; k += blockDim.x) eg[dd + k] = M0[k * d + k];
            }
            break;
        }
        case OP_EIGMAX:
        case OP_EIGMIN: {  // out[i] = extreme eigenvalue, out2 = v v^T (its Euclidean gradient w.r.t. the symmetric matrix)
            lds_load(A, M0, d);
            lds_symmetrize(M0, M5, d);
            lds_eigh<QL>(M0, M1, cs, d);
            int best = 0;
            for (int k = 1; k < d; ++k) {
                bool better = op == OP_EIGMAX ? (M0[k * d + k] > M0[best * d + best]) : (M0[k * d + k] < M0[best * d + best]);
                if (better) best = k;
            }
            if (threadIdx.x == 0) out[i] = M0[best * d + best];
            if (out2) {
                for (int k = threadIdx.x; k < dd; k += blockDim.x) {
                    int r = k / d, cc = k - r * d;
                    out2[i * dd + k] = M1[r * d + best] * M1[cc * d + best];
                }
            }
            break;
        }
        default: break;
    }
    if (!ok && threadIdx.x == 0 && status) {
        if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)i;
    }
}

// Y = W^T X W straight from / to Mandel vectors: x (n x D_vec), w (D x dl), y (n x dl_vec).   nested_spd_utils.py:13-48
__global__ __launch_bounds__(64) void spd_project_kernel(const double* __restrict__ x, const double* __restrict__ w,
                                                         double* __restrict__ y, int64_t n, int D, int dl) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* X = lds;              // D x D
    double* Wl = X + D * D;       // D x dl
    double* T = Wl + D * dl;      // D x dl : X W
    const int64_t i = blockIdx.x;
    lds_from_mandel(x + i * (int64_t)(D * (D + 1) / 2), X, D);
    for (int e = threadIdx.x; e < D * dl; e += blockDim.x) Wl[e] = w[e];
    wsync();
    for (int e = threadIdx.x; e < D * dl; e += blockDim.x) {
        int r = e / dl, c = e - r * dl;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s = __builtin_fma(X[r * D + k], Wl[k * dl + c], s);
        T[e] = s;
    }
    wsync();
    const int dv = dl * (dl + 1) / 2;
    for (int e = threadIdx.x; e < dv; e += blockDim.x) {
        int k = 0;
        while (k + 1 < dl && (k + 1) * dl - (k + 1) * k / 2 <= e) ++k;
        int c = e - (k * dl - k * (k - 1) / 2);
        int r = c + k;
        double s1 = 0.0, s2 = 0.0;   // (r,c) and (c,r): averaged like symmetric_matrix_to_vector_mandel_torch
        for (int q = 0; q < D; ++q) {
            s1 = __builtin_fma(Wl[q * dl + r], T[q * dl + c], s1);
            s2 = __builtin_fma(Wl[q * dl + c], T[q * dl + r], s2);
        }
        y[i * dv + e] = (k == 0) ? s1 : 0.5 * (kSqrt2 * s1 + kSqrt2 * s2);
    }
}

// The same projection for a small latent dimension (the nested kernels: dl = 2, 3, 4) as ONE matrix-vector product in Mandel coordinates:
// y = P x with P (dl_vec x D_vec) a function of W only,
//     y_ab = sum_r W_ra W_rb X_rr + sum_{r > c} (W_ra W_cb + W_ca W_rb) X_rc,     X_rc = x_k / sqrt2 (r != c),   y_e = sqrt2 y_ab (a != b).
// Every block builds P in LDS (D_vec x dl_vec products: less than one matrix's worth of the old kernel's work), then each WAVE does one
// matrix per step: a coalesced read of the D_vec entries, dl_vec FMAs per entry and dl_vec wave reductions.  The wave-per-matrix LDS
// kernel above spends its time in barrier phases whose last one keeps dl_vec of 64 lanes busy: at n = 4096, D = 20 -> 2 a call
// went from 16.5 to 13.6 us host-timed (most of it launch overhead); rocprofv3 kernel time of this kernel 9 us (a 4096-element fill kernel: 3-5 us).
template <int DL>
__global__ __launch_bounds__(256) void spd_project_small_kernel(const double* __restrict__ x, const double* __restrict__ w,
                                                                double* __restrict__ y, int64_t n, int D) {
    constexpr int DV = DL * (DL + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Dv = D * (D + 1) / 2;
    double* Wl = lds;                 // D x DL
    double* P = Wl + D * DL;          // DV x Dv
    build_projection_operator<DL>(w, D, Wl, P);
    const int lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    for (int64_t i = (int64_t)blockIdx.x * waves + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * waves) {
        double acc[DV];
        static_for<DV>([&](auto e) { acc[decltype(e)::value] = 0.0; });
        const double* xi = x + i * Dv;
        for (int k = lane; k < Dv; k += 64) {
            const double xk = xi[k];
            static_for<DV>([&](auto e) { acc[decltype(e)::value] = __builtin_fma(P[decltype(e)::value * Dv + k], xk, acc[decltype(e)::value]); });
        }
        static_for<DV>([&](auto e) {
            double s = acc[decltype(e)::value];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == decltype(e)::value) y[i * DV + decltype(e)::value] = s;
        });
    }
}

// logm for d <= 8, one LANE per matrix: register eigen-decomposition (spd_eigvec.hpp), y = V diag(log lambda) V^T.  The wave-per-matrix
// LDS Jacobi kernel below takes 13 us for 4096 2 x 2 matrices (64 blocks' worth of useful lanes spread over 4096 blocks).
template <int D>
__global__ __launch_bounds__(64) void spd_logm_mandel_reg_kernel(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    constexpr int T = tri_size(D);
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t ic = i < n ? i : n - 1;             // every lane of the wave runs the eigen-solver (it leaves its stages by a wave vote)
    double m[T], lam[D], v[D * D];
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const double e = x[ic * T + mandel_pos(D, r, c)];
            m[tri(r, c)] = (r == c) ? e : e / kSqrt2;
        });
    });
    sym_eig_reg<D>(m, lam, v);
    double lg[D];
    static_for<D>([&](auto kk) { lg[decltype(kk)::value] = log(lam[decltype(kk)::value]); });     // (OCML: NaN for a non-positive eigenvalue, like the reference)
    if (i >= n) return;
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            double s = 0.0;
            static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; s = __builtin_fma(v[r * D + k] * lg[k], v[c * D + k], s); });
            y[i * T + mandel_pos(D, r, c)] = (r == c) ? s : kSqrt2 * s;
        });
    });
}

// logm of SPD matrices, Mandel in -> Mandel out (the per-point part of SpdLogEuclideanGaussianKernel, kernels_spd.py:289-305)
template <bool QL>
__global__ __launch_bounds__(64) void spd_logm_mandel_kernel(const double* __restrict__ x, double* __restrict__ y, int64_t n, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* cs = M2 + dd;
    const int64_t i = blockIdx.x;
    const int dv = d * (d + 1) / 2;
    lds_from_mandel(x + i * dv, M0, d);
    lds_eigh<QL>(M0, M1, cs, d);
    lds_fun_from_eig(M0, M1, M2, d, FN_LOG, cs);
    for (int e = threadIdx.x; e < dv; e += blockDim.x) {
        int k = 0;
        while (k + 1 < d && (k + 1) * d - (k + 1) * k / 2 <= e) ++k;
        int c = e - (k * d - k * (k - 1) / 2);
        int r = c + k;
        y[i * dv + e] = (k == 0) ? M2[r * d + c] : 0.5 * (kSqrt2 * M2[r * d + c] + kSqrt2 * M2[c * d + r]);
    }
}

// Pairwise Frobenius distance between symmetric matrices given as Mandel vectors, with the reference's +1e-15 on EVERY
// matrix element of the difference (spd_utils_torch.py:156): in Mandel coordinates that is +1e-15 on diagonal entries and
// +sqrt2*1e-15 on off-diagonal ones.  out = d, exp(-beta d^2) or exp(-beta d).
// grid.x = (row chunk, column group), column group fastest; grid.y = batch.  Lane = column j: its x2 vector stays in registers for all
// `rows` rows of the block when it is short (DV_REG > 0: d <= 3, the latent spaces of the nested kernels), the x1 row is wave-uniform
// (scalar loads); the fp64 stores of a wave are one contiguous 512-byte run.  8 B/pair of HBM traffic (the output): write-bound once the
// per-output arithmetic is below ~45 instructions - the first version (one thread per output with two 64-bit divisions, 85 us at
// N = 4096, d = 2) was 3.3x off the 26 us write floor.  Gaussian mode skips the square root: exp(-beta s) with s = d^2.
template <int DSMALL>        // DSMALL = d for d <= 3 (x2 vector in registers, compile-time length), 0 = any d
__global__ __launch_bounds__(256) void frobenius_pairwise_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                                 double* __restrict__ out, int64_t n1, int64_t n2, int d, int64_t s1,
                                                                 int64_t s2, int col_blocks, int rows, double beta, int flags) {
    constexpr int DV_REG = DSMALL * (DSMALL + 1) / 2;
    const int dv = DSMALL > 0 ? DV_REG : d * (d + 1) / 2;
    const int mode = flags & GABO_OUT_MASK;
    __shared__ double tab[64];                              // 2^(j/64) for exp_neg_tab
    if (threadIdx.x < 64) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
    __syncthreads();
    double ec[6];
    static_for<6>([&](auto k) { ec[decltype(k)::value] = kExpTabC[decltype(k)::value]; });
    double ec3 = ec[3];
    asm volatile("" : "+v"(ec3));
    const double neg_beta = -beta;
    const uint32_t rc = blockIdx.x / (uint32_t)col_blocks, cg = blockIdx.x - rc * (uint32_t)col_blocks;
    const int64_t b = blockIdx.y;
    const int64_t j = (int64_t)cg * blockDim.x + threadIdx.x;
    const int64_t jc = j < n2 ? j : n2 - 1;              // out-of-range lanes recompute the last column and do not store
    const int64_t i0 = (int64_t)rc * rows;
    const int64_t i1 = i0 + rows < n1 ? i0 + rows : n1;
    const double* q = x2 + b * s2 + jc * dv;
    double qv[DV_REG > 0 ? DV_REG : 1];
    if constexpr (DV_REG > 0) {
        static_for<DV_REG>([&](auto ee) { qv[decltype(ee)::value] = q[decltype(ee)::value]; });
    }
    double* ob = out + b * n1 * n2 + j;
    for (int64_t i = i0; i < i1; ++i) {
        const double* p = x1 + b * s1 + i * dv;           // wave-uniform
        double s = 0.0;
        if constexpr (DV_REG > 0) {
            static_for<DV_REG>([&](auto ee) {
                constexpr int e = decltype(ee)::value;
                double diff = (p[e] - qv[e]) + (e < DSMALL ? 1e-15 : kSqrt2 * 1e-15);
                s = __builtin_fma(diff, diff, s);
            });
        } else {
            for (int e = 0; e < dv; ++e) {
                double diff = (p[e] - q[e]) + (e < d ? 1e-15 : kSqrt2 * 1e-15);
                s = __builtin_fma(diff, diff, s);
            }
        }
        double val;
        if (mode == GABO_OUT_GAUSSIAN) {
            val = exp_neg_tab(s * neg_beta, ec, ec3, tab);  // sqrt(s)^2 = s to an ulp (kernels_spd.py:238-240, 309-311)
        } else {
            const double dist = __builtin_sqrt(s);
            val = mode == GABO_OUT_DISTANCE ? dist : exp_neg_tab(dist * neg_beta, ec, ec3, tab);
        }
        // (the argument clamp of the table exp returns its bound for a NaN; a NaN entry - e.g. the logm of a matrix with a non-positive
        // eigenvalue in the log-Euclidean kernels - must give a NaN kernel value, as torch.exp does in kernels_spd.py:238-240, 309-311)
        val = s != s ? s : val;
        if (j < n2) ob[i * n2] = val;
    }
}

// Adjoint of the Frechet derivative of logm (Daleckii-Krein): gx = V ((V^T G V) o F) V^T, F_kl = (log l_k - log l_l)/(l_k - l_l),
// F_kk = 1/l_k, Mandel in / out.  The Mandel map is an isometry, so gx is the gradient of the loss w.r.t. the Mandel
// vector x when g is its gradient w.r.t. the Mandel vector of logm(X): what autograd through logm_torch
// (spd_utils_torch.py:13-30) produces, without its 1/(l_k - l_l) singularity at repeated eigenvalues.
template <bool QL>
__global__ __launch_bounds__(64) void spd_logm_mandel_backward_kernel(const double* __restrict__ x, const double* __restrict__ g,
                                                                      double* __restrict__ gx, int64_t n, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* cs = M3 + dd;
    const int64_t i = blockIdx.x;
    const int dv = d * (d + 1) / 2;
    lds_from_mandel(x + i * dv, M0, d);
    lds_from_mandel(g + i * dv, M2, d);
    lds_eigh<QL>(M0, M1, cs, d);                  // M0 = diag(lambda), M1 = V
    lds_mm(M1, M2, M3, d, true, false);         // V^T G
    lds_mm(M3, M1, M2, d, false, false);        // V^T G V
    for (int k = threadIdx.x; k < d; k += blockDim.x) cs[k] = log(M0[k * d + k]);     // (d logarithms, not 2 d^2: the eigen-solver's scratch is free)
    wsync();
    for (int e = threadIdx.x; e < dd; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        double lr = M0[r * d + r], lc = M0[c * d + c];
        double mean = 0.5 * (lr + lc), dl = lr - lc;
        // log(lr/lc)/(lr-lc) = atanh(z)/(z mean) with z = dl/(2 mean): series near z = 0 keeps full precision
        double z = dl / (2.0 * mean), z2 = z * z;
        double f = (__builtin_fabs(z) < 1e-3) ? (1.0 + z2 * (1.0 / 3.0 + z2 * (0.2 + z2 * (1.0 / 7.0)))) / mean
                                              : (cs[r] - cs[c]) / dl;
        M2[e] *= f;
    }
    wsync();
    lds_mm(M1, M2, M3, d, false, false);        // V (.)
    lds_mm(M3, M1, M2, d, false, true);         // V (.) V^T
    for (int e = threadIdx.x; e < dv; e += blockDim.x) {
        int k = 0;
        while (k + 1 < d && (k + 1) * d - (k + 1) * k / 2 <= e) ++k;
        int c = e - (k * d - k * (k - 1) / 2);
        int r = c + k;
        gx[i * dv + e] = (k == 0) ? M2[r * d + c] : 0.5 * (kSqrt2 * M2[r * d + c] + kSqrt2 * M2[c * d + r]);
    }
}

// Adjoint of the Frechet derivative of a primary matrix function f in {log, exp, sqrt} at the symmetric matrix A, matrices in and
// out: out = V ((V^T sym(G) V) o F) V^T with F_kl the divided differences of f at the eigenvalues (F_kk = f').  This is what autograd
// through logm_torch / sqrtm_torch (spd_utils_torch.py:13-50) computes, in the form that stays finite at repeated eigenvalues.
// eig != nullptr: the eigen-decomposition saved by the forward launch (n x (d^2 + d): V, then the eigenvalues) replaces the Jacobi
// solve of `a`, which is most of this kernel's time.
template <int THREADS, bool QL>
__global__ __launch_bounds__(THREADS) void spd_matfun_backward_kernel(const double* __restrict__ a, const double* __restrict__ eig,
                                                                 const double* __restrict__ g, double* __restrict__ out, int64_t n,
                                                                 int d, int fn) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* cs = M3 + dd;
    const int64_t i = blockIdx.x;
    lds_load(g + i * dd, M2, d);
    lds_symmetrize(M2, M3, d);
    if (eig) {
        const double* eg = eig + i * (dd + d);
        for (int k = threadIdx.x; k < dd; k += blockDim.x) M1[k] = eg[k];
        for (int k = threadIdx.x; k < d; k += blockDim.x) M0[k * d + k] = eg[dd + k];
        wsync();
    } else {
        lds_load(a + i * dd, M0, d);
        lds_symmetrize(M0, M3, d);
        lds_eigh<QL>(M0, M1, cs, d);              // M0 = diag(lambda), M1 = V
    }
    lds_mm(M1, M2, M3, d, true, false);         // V^T G
    lds_mm(M3, M1, M2, d, false, false);        // V^T G V
    for (int k = threadIdx.x; k < d; k += blockDim.x) {                       // f(lambda_k) once per eigenvalue (the eigen-solver's scratch is free)
        const double lam = M0[k * d + k];
        cs[k] = fn == FN_LOG ? log(lam) : (fn == FN_SQRT ? __builtin_sqrt(lam) : 0.0);
    }
    wsync();
    for (int e = threadIdx.x; e < dd; e += blockDim.x) {
        int r = e / d, c = e - r * d;
        const double lr = M0[r * d + r], lc = M0[c * d + c];
        const double mean = 0.5 * (lr + lc), dl = lr - lc;
        double f;
        if (fn == FN_LOG) {
            const double z = dl / (2.0 * mean), z2 = z * z;
            f = (__builtin_fabs(z) < 1e-3) ? (1.0 + z2 * (1.0 / 3.0 + z2 * (0.2 + z2 * (1.0 / 7.0)))) / mean : (cs[r] - cs[c]) / dl;
        } else if (fn == FN_SQRT) {
            f = 1.0 / (cs[r] + cs[c]);                                      // (sqrt lr - sqrt lc)/(lr - lc), exact and stable
        } else {
            const double h = 0.5 * dl, h2 = h * h;                         // (e^lr - e^lc)/(lr - lc) = e^mean sinh(h)/h
            const double sh = (__builtin_fabs(h) < 1e-2) ? 1.0 + h2 * (1.0 / 6.0 + h2 * (1.0 / 120.0 + h2 / 5040.0)) : sinh(h) / h;
            f = exp(mean) * sh;
        }
        M2[e] *= f;
    }
    wsync();
    lds_mm(M1, M2, M3, d, false, false);        // V (.)
    lds_mm(M3, M1, M2, d, false, true);         // V (.) V^T
    lds_symmetrize(M2, M3, d);
    lds_store(M2, out + i * dd, d);
}

// Gradient of frobenius_pairwise w.r.t. x1 (Mandel): gx1[b,i,e] = sum_j w_ij (x1_ie - x2_je + sgn*eps_e), with
// w_ij = go_ij * dOut/d(d^2) * 2 recomputed from the inputs (Gaussian: -2 beta K; Laplace: -beta K / d; distance: 1/d).
// One block per (b, i): phase 1 the threads own 256 columns j and put w_j in LDS, phase 2 they own the Mandel entries e and
// walk the 256 columns with coalesced reads of x2.  The x2-gradient is this kernel with the sets exchanged and eps_sign = -1.
__global__ __launch_bounds__(256) void frobenius_backward_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                                 const double* __restrict__ go, double* __restrict__ gx1,
                                                                 int64_t n1, int64_t n2, int d, int64_t s1, int64_t s2,
                                                                 int64_t go_bs, int64_t go_rs, int64_t go_cs, double beta, int flags,
                                                                 double eps_sign) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int dv = d * (d + 1) / 2;
    double* P = lds;          // dv : x1 row + eps
    double* Wj = P + dv;      // 256
    const int mode = flags & GABO_OUT_MASK;
    const int64_t b = blockIdx.x / n1, i = blockIdx.x - b * n1;
    const double* p = x1 + b * s1 + i * dv;
    for (int e = threadIdx.x; e < dv; e += blockDim.x) P[e] = p[e] + eps_sign * (e < d ? 1e-15 : kSqrt2 * 1e-15);
    __syncthreads();
    const int per = (dv + (int)blockDim.x - 1) / (int)blockDim.x;   // <= 3 for d <= 32
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t j0 = 0; j0 < n2; j0 += blockDim.x) {
        int64_t j = j0 + threadIdx.x;
        double w = 0.0;
        if (j < n2) {
            const double* q = x2 + b * s2 + j * dv;
            double s = 0.0;
            for (int e = 0; e < dv; ++e) { double diff = P[e] - q[e]; s = __builtin_fma(diff, diff, s); }
            double g = go[b * go_bs + i * go_rs + j * go_cs];
            if (mode == GABO_OUT_GAUSSIAN) w = g * (-2.0 * beta) * exp(-(s * beta));
            else {
                double dist = __builtin_sqrt(s);
                w = mode == GABO_OUT_LAPLACE ? g * (-beta) * exp(-(dist * beta)) / dist : g / dist;
            }
        }
        Wj[threadIdx.x] = w;
        __syncthreads();
        int64_t cnt = n2 - j0 < (int64_t)blockDim.x ? n2 - j0 : (int64_t)blockDim.x;
        for (int t = 0; t < per; ++t) {
            int e = threadIdx.x + t * blockDim.x;
            if (e < dv) {
                const double* q = x2 + b * s2 + j0 * dv + e;
                double pe = P[e], a = acc[t];
                for (int64_t jj = 0; jj < cnt; ++jj) a = __builtin_fma(Wj[jj], pe - q[jj * dv], a);
                acc[t] = a;
            }
        }
        __syncthreads();
    }
    for (int t = 0; t < per; ++t) {
        int e = threadIdx.x + t * blockDim.x;
        if (e < dv) gx1[(b * n1 + i) * dv + e] = acc[t];
    }
}

// Sphere manifold operations, one lane per point (dim is small): x, u, v: n x dim.
enum { SOP_PROJ = 0, SOP_RETR = 1, SOP_EXP = 2, SOP_LOG = 3, SOP_DIST = 4, SOP_EHESS2RHESS = 5 };

__global__ __launch_bounds__(256) void sphere_manifold_kernel(int op, const double* __restrict__ x, const double* __restrict__ u,
                                                              const double* __restrict__ v, const double* __restrict__ w,
                                                              double* __restrict__ out, int64_t n, int dim) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* X = x + i * dim;
    const double* U = u + i * dim;
    double* O = out + i * dim;
    switch (op) {
        case SOP_PROJ: {  // H - <X,H> X                          [3P] Sphere.proj / egrad2rgrad / transp(.,Y,U) = proj(Y,U)
            double ip = 0.0;
            for (int k = 0; k < dim; ++k) ip = __builtin_fma(X[k], U[k], ip);
            for (int k = 0; k < dim; ++k) O[k] = U[k] - ip * X[k];
            break;
        }
        case SOP_RETR: {  // (X+U)/|X+U|                           [3P] Sphere.retr
            double nn = 0.0;
            for (int k = 0; k < dim; ++k) { double y = X[k] + U[k]; nn = __builtin_fma(y, y, nn); }
            double inv = 1.0 / __builtin_sqrt(nn);
            for (int k = 0; k < dim; ++k) O[k] = (X[k] + U[k]) * inv;
            break;
        }
        case SOP_EXP: {  // x cos|u| + u sin|u|/|u| ; x where |u| < 1e-16     sphere_utils.py:14-38 (base = x, tangent = u)
            double nn = 0.0;
            for (int k = 0; k < dim; ++k) nn = __builtin_fma(U[k], U[k], nn);
            double nu = __builtin_sqrt(nn);
            if (nu < 1e-16) {
                for (int k = 0; k < dim; ++k) O[k] = X[k];
            } else {
                double cn = cos(nu), sn = sin(nu) / nu;
                for (int k = 0; k < dim; ++k) O[k] = X[k] * cn + U[k] * sn;
            }
            break;
        }
        case SOP_LOG: {  // (y - x cos t) t / sin t, t = acos(clip(<x,y>)); 0 where t < 1e-16   sphere_utils.py:41-65 (base = x, point = u)
            double ip = 0.0;
            for (int k = 0; k < dim; ++k) ip = __builtin_fma(X[k], U[k], ip);
            ip = ip > 1.0 ? 1.0 : (ip < -1.0 ? -1.0 : ip);
            double th = acos(ip);
            if (th < 1e-16) {
                for (int k = 0; k < dim; ++k) O[k] = 0.0;
            } else {
                double cn = cos(th), f = th / sin(th);
                for (int k = 0; k < dim; ++k) O[k] = (U[k] - X[k] * cn) * f;
            }
            break;
        }
        case SOP_DIST: {  // acos(clip(<x,y>, -1, 1))              [3P] Sphere.dist
            double ip = 0.0;
            for (int k = 0; k < dim; ++k) ip = __builtin_fma(X[k], U[k], ip);
            ip = ip > 1.0 ? 1.0 : (ip < -1.0 ? -1.0 : ip);
            out[i] = acos(ip);
            break;
        }
        case SOP_EHESS2RHESS: {  // proj(x, eh) - <x, eg> u : u = egrad, v = ehess, w = tangent    [3P]
            const double* EH = v + i * dim;
            const double* TG = w + i * dim;
            double a = 0.0, bq = 0.0;
            for (int k = 0; k < dim; ++k) { a = __builtin_fma(X[k], EH[k], a); bq = __builtin_fma(X[k], U[k], bq); }
            for (int k = 0; k < dim; ++k) O[k] = EH[k] - a * X[k] - bq * TG[k];
            break;
        }
        default: break;
    }
}

}  // namespace gabo

extern "C" {

int gabo_spd_manifold_op(int op, const double* a, const double* b, const double* c, const double* e, double* out, double* out2,
                         int64_t n, int d, int* status, gabo_stream_t stream) {
    if (n < 0 || op < 0 || op > gabo::OP_EIGMIN) return GABO_ERR_ARG;
    if (d < 1 || d > 32) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    if (!a || !out) return GABO_ERR_ARG;
    const bool need_b = op <= gabo::OP_EHESS2RHESS;
    if (need_b && !b) return GABO_ERR_ARG;
    if ((op == gabo::OP_INNER || op == gabo::OP_EHESS2RHESS) && !c) return GABO_ERR_ARG;
    if (op == gabo::OP_EHESS2RHESS && !e) return GABO_ERR_ARG;
    if (n > 0x7fffffffLL) return GABO_ERR_ARG;
    size_t lds = (size_t)(6 * d * d + gabo::kJacobiScratch) * sizeof(double);
    // four waves per matrix only share the O(d^3) products (the eigen-solve is one wave's): worth it for a few matrices, not for thousands
    if (d > 12 && n < 1024)
        hipLaunchKernelGGL((gabo::spd_manifold_kernel<256, true>), dim3((unsigned)n), dim3(256), lds, (hipStream_t)stream, op, a, b, c, e, out, out2, n,
                       d, status);
    else if (d >= gabo::kWaveEighMinDim)
        hipLaunchKernelGGL((gabo::spd_manifold_kernel<64, true>), dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, op, a, b, c, e, out, out2, n,
                       d, status);
    else
        hipLaunchKernelGGL((gabo::spd_manifold_kernel<64, false>), dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, op, a, b, c, e, out, out2, n,
                       d, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_project(const double* x_mandel, const double* w, double* y_mandel, int64_t n, int D, int dl, gabo_stream_t stream) {
    if (n < 0) return GABO_ERR_ARG;
    if (D < 1 || D > 64 || dl < 1 || dl > 64) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    if (!x_mandel || !w || !y_mandel || n > 0x7fffffffLL) return GABO_ERR_ARG;
    const size_t small_lds = (size_t)(D * dl + (dl * (dl + 1) / 2) * (D * (D + 1) / 2)) * sizeof(double);
    if (dl >= 2 && dl <= 4 && small_lds <= 48 * 1024) {
        const int64_t blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;      // one matrix per wave (two per wave measured slower: 10.6 vs 9.0 us at n = 4096)
        switch (dl) {
            case 2: hipLaunchKernelGGL(gabo::spd_project_small_kernel<2>, dim3((unsigned)blocks), dim3(256), small_lds, (hipStream_t)stream, x_mandel, w, y_mandel, n, D); break;
            case 3: hipLaunchKernelGGL(gabo::spd_project_small_kernel<3>, dim3((unsigned)blocks), dim3(256), small_lds, (hipStream_t)stream, x_mandel, w, y_mandel, n, D); break;
            default: hipLaunchKernelGGL(gabo::spd_project_small_kernel<4>, dim3((unsigned)blocks), dim3(256), small_lds, (hipStream_t)stream, x_mandel, w, y_mandel, n, D); break;
        }
        return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
    }
    size_t lds = (size_t)(D * D + 2 * D * dl) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_project_kernel, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x_mandel, w, y_mandel, n, D, dl);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_logm_mandel(const double* x_mandel, double* y_mandel, int64_t n, int d, gabo_stream_t stream) {
    if (n < 0) return GABO_ERR_ARG;
    if (d < 1 || d > 32) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    if (!x_mandel || !y_mandel || n > 0x7fffffffLL) return GABO_ERR_ARG;
    if (d >= 2 && d <= 8) {
        const unsigned blocks = (unsigned)((n + 63) / 64);
#define GABO_CASE(DD) \
    case DD: hipLaunchKernelGGL(gabo::spd_logm_mandel_reg_kernel<DD>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, x_mandel, y_mandel, n); break;
        switch (d) { GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) }
#undef GABO_CASE
        return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
    }
    size_t lds = (size_t)(3 * d * d + gabo::kJacobiScratch) * sizeof(double);
    if (d >= gabo::kWaveEighMinDim)
        hipLaunchKernelGGL(gabo::spd_logm_mandel_kernel<true>, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x_mandel, y_mandel, n, d);
    else
        hipLaunchKernelGGL(gabo::spd_logm_mandel_kernel<false>, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x_mandel, y_mandel, n, d);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_frobenius_pairwise(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2, int d,
                            int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags, gabo_stream_t stream) {
    if (batch < 0 || n1 < 0 || n2 < 0 || x1_batch_stride < 0 || x2_batch_stride < 0) return GABO_ERR_ARG;
    if (d < 1 || d > 64) return GABO_ERR_DIM;
    if (batch == 0 || n1 == 0 || n2 == 0) return GABO_OK;
    if (!x1 || !x2 || !out) return GABO_ERR_ARG;
    const int threads = n2 >= 256 ? 256 : (n2 > 128 ? 192 : (n2 > 64 ? 128 : 64));
    const int64_t col_blocks = (n2 + threads - 1) / threads;
    int rows = 64;                                    // rows per block: as many as keep >= ~2048 blocks in flight
    while (rows > 1 && col_blocks * ((n1 + rows - 1) / rows) * batch < 2048) rows >>= 1;
    const int64_t blocks = col_blocks * ((n1 + rows - 1) / rows);
    if (blocks > 0x7fffffffLL) return GABO_ERR_ARG;
    // the batch rides in grid.y (<= 65535): larger t-batches (raw_samples x 1 x d candidates against one training set) go out in slices
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        const dim3 grid((unsigned)blocks, (unsigned)nb);
        const double* x1b = x1 + b0 * x1_batch_stride;
        const double* x2b = x2 + b0 * x2_batch_stride;
        double* outb = out + b0 * n1 * n2;
#define GABO_FROB_LAUNCH(DS)                                                                                                       \
    hipLaunchKernelGGL(gabo::frobenius_pairwise_kernel<DS>, grid, dim3(threads), 0, (hipStream_t)stream, x1b, x2b, outb, n1, n2, d, \
                       x1_batch_stride, x2_batch_stride, (int)col_blocks, rows, beta, flags)
        if (d == 1) GABO_FROB_LAUNCH(1);
        else if (d == 2) GABO_FROB_LAUNCH(2);
        else if (d == 3) GABO_FROB_LAUNCH(3);
        else GABO_FROB_LAUNCH(0);
    }
#undef GABO_FROB_LAUNCH
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_logm_mandel_backward(const double* x_mandel, const double* grad_y, double* grad_x, int64_t n, int d,
                                  gabo_stream_t stream) {
    if (d < 1 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (n < 0 || (n > 0 && (!x_mandel || !grad_y || !grad_x))) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    size_t lds = (size_t)(4 * d * d + gabo::kJacobiScratch) * sizeof(double);
    if (d >= gabo::kWaveEighMinDim)
        hipLaunchKernelGGL(gabo::spd_logm_mandel_backward_kernel<true>, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x_mandel,
                           grad_y, grad_x, n, d);
    else
        hipLaunchKernelGGL(gabo::spd_logm_mandel_backward_kernel<false>, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x_mandel,
                           grad_y, grad_x, n, d);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_matfun_backward(int op, const double* a, const double* grad_out, double* grad_a, int64_t n, int d, gabo_stream_t stream) {
    if (d < 1 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (op != GABO_SPD_LOGM && op != GABO_SPD_EXPM && op != GABO_SPD_SQRTM) return GABO_ERR_ARG;
    if (n < 0 || (n > 0 && (!a || !grad_out || !grad_a))) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    const int fn = op == GABO_SPD_LOGM ? gabo::FN_LOG : (op == GABO_SPD_EXPM ? gabo::FN_EXP : gabo::FN_SQRT);
    size_t lds = (size_t)(4 * d * d + gabo::kJacobiScratch) * sizeof(double);
    if (d > 12 && n < 1024)
        hipLaunchKernelGGL((gabo::spd_matfun_backward_kernel<256, true>), dim3((unsigned)n), dim3(256), lds, (hipStream_t)stream, a, (const double*)nullptr,
                       grad_out, grad_a, n, d, fn);
    else if (d >= gabo::kWaveEighMinDim)
        hipLaunchKernelGGL((gabo::spd_matfun_backward_kernel<64, true>), dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, a, (const double*)nullptr,
                       grad_out, grad_a, n, d, fn);
    else
        hipLaunchKernelGGL((gabo::spd_matfun_backward_kernel<64, false>), dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, a, (const double*)nullptr,
                       grad_out, grad_a, n, d, fn);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_matfun_backward_eig(int op, const double* eig, const double* grad_out, double* grad_a, int64_t n, int d,
                                 gabo_stream_t stream) {
    if (d < 1 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (op != GABO_SPD_LOGM && op != GABO_SPD_EXPM && op != GABO_SPD_SQRTM) return GABO_ERR_ARG;
    if (n < 0 || (n > 0 && (!eig || !grad_out || !grad_a))) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    const int fn = op == GABO_SPD_LOGM ? gabo::FN_LOG : (op == GABO_SPD_EXPM ? gabo::FN_EXP : gabo::FN_SQRT);
    size_t lds = (size_t)(4 * d * d + gabo::kJacobiScratch) * sizeof(double);
    // (the eigen-decomposition is handed over: no solver in this launch, the Jacobi instantiation is the small one)
    if (d > 12)
        hipLaunchKernelGGL((gabo::spd_matfun_backward_kernel<256, false>), dim3((unsigned)n), dim3(256), lds, (hipStream_t)stream, (const double*)nullptr,
                       eig, grad_out, grad_a, n, d, fn);
    else
        hipLaunchKernelGGL((gabo::spd_matfun_backward_kernel<64, false>), dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, (const double*)nullptr,
                       eig, grad_out, grad_a, n, d, fn);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_frobenius_backward(const double* x1, const double* x2, const double* grad_out, double* grad_x1, int64_t batch, int64_t n1,
                            int64_t n2, int d, int64_t x1_batch_stride, int64_t x2_batch_stride, int64_t go_batch_stride,
                            int64_t go_row_stride, int64_t go_col_stride, double beta, int flags, double eps_sign,
                            gabo_stream_t stream) {
    if (d < 1 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (batch < 0 || n1 < 0 || n2 < 0 || (flags & ~GABO_OUT_MASK)) return GABO_ERR_ARG;
    if (batch * n1 == 0) return GABO_OK;
    if (!x1 || !grad_x1 || (n2 > 0 && (!x2 || !grad_out))) return GABO_ERR_ARG;
    size_t lds = (size_t)(d * (d + 1) / 2 + 256) * sizeof(double);
    hipLaunchKernelGGL(gabo::frobenius_backward_kernel, dim3((unsigned)(batch * n1)), dim3(256), lds, (hipStream_t)stream, x1, x2,
                       grad_out, grad_x1, n1, n2, d, x1_batch_stride, x2_batch_stride, go_batch_stride, go_row_stride,
                       go_col_stride, beta, flags, eps_sign);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_sphere_manifold_op(int op, const double* x, const double* u, const double* v, const double* w, double* out, int64_t n,
                            int dim, gabo_stream_t stream) {
    if (n < 0 || op < 0 || op > gabo::SOP_EHESS2RHESS) return GABO_ERR_ARG;
    if (dim < 1) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    if (!x || !u || !out) return GABO_ERR_ARG;
    if (op == gabo::SOP_EHESS2RHESS && (!v || !w)) return GABO_ERR_ARG;
    hipLaunchKernelGGL(gabo::sphere_manifold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, x, u, v,
                       w, out, n, dim);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
}
