// Gradient of the SPD affine-invariant pairwise kernels with respect to the FIRST argument (Mandel vectors).
//
// The reference gets this by autograd through cholesky / inverse / bmm / symeig(eigenvectors=True) / log / exp
// (Riemannian_utils/spd_utils_torch.py:87-120, kernel_utils/kernels_spd.py:91-100).  Closed form:
//
//     d(d_ij^2)/dA_i = -2 L_i^-T logm(M_ij) L_i^-1,      M_ij = L_i^-1 B_j L_i^-T,  L_i = chol(A_i)
//     dLoss/dA_i     = -2 L_i^-T [ sum_j w_ij logm(M_ij) ] L_i^-1,   w_ij = dLoss/d(d_ij^2)
//
// so only the matrix FUNCTION logm(M) is needed - never an eigenvector derivative - and the congruence with L_i^-1 is
// applied once per row i.  The gradient with respect to the second argument is this same kernel with the roles of
// the two sets exchanged and grad_out read transposed (d is symmetric in its arguments).
//
// Mapping: one wave per (batch, i); lane = column j, looping over j in chunks of 64.  Each lane forms M (as in the
// forward kernel), diagonalises it WITH eigenvectors, builds logm(M) = V diag(log lambda) V^T, and accumulates
// w_ij logm(M_ij) into its own LDS column; one cross-lane reduction per row, then the lanes share the final congruence
// and the Mandel scatter.  Eigen-solver: Householder + implicit QL with the eigenvector matrix in registers (spd_eigvec.hpp);
// Z takes 200 VGPRs at d = 10, so above d = 8 the kernel is budgeted for ONE wave per SIMD and 512 VGPRs.  The first version
// ran cyclic Jacobi (V in registers up to d = 8, in LDS above: 51-74 KB per wave, ~4e4 instructions per pair); full N = 4096
// backward, ms, Jacobi -> QL: d = 8: 39.7 -> see DESIGN.md, d = 10: 70 -> 8.8, d = 12: 228 -> 23.
#include "gabo_device.hpp"
#include "spd_prep.hpp"
#include "spd_jacobi.hpp"
#include "spd_eigvec.hpp"
#include "spd_generic.hpp"
#include "../../include/gabo_hip.h"

#ifndef GABO_BWD_JACOBI_MAX_DIM
#define GABO_BWD_JACOBI_MAX_DIM 0   /* dimensions up to this use the cyclic Jacobi (A/B builds); the QL solver is faster from d = 3 on */
#endif
#ifndef GABO_BWD_TWO_WAVE_MAX_DIM
#define GABO_BWD_TWO_WAVE_MAX_DIM 8  /* Z + M fit 256 VGPRs up to here: two waves per SIMD */
#endif

namespace gabo {

template <int D>
__global__ __launch_bounds__(64, (D > GABO_BWD_TWO_WAVE_MAX_DIM ? 1 : 2)) void spd_ai_backward_kernel(const double* __restrict__ Winv, const double* __restrict__ G,
                                                             const double* __restrict__ gout, double* __restrict__ gx,
                                                             int64_t n1, int64_t n2, int64_t w_batch_stride,
                                                             int64_t g_batch_stride, int64_t go_sb, int64_t go_si, int64_t go_sj,
                                                             double beta, int flags) {
    constexpr int T = tri_size(D);
    // accumulators of sum_j w_ij logm(M_ij): one LDS column per lane (T x 64 doubles: 28 KB at d = 10, 40 KB at d = 12; with one
    // wave per SIMD that is at most 4 blocks = 160 KB per CU), reduced across the wave once per row
    __shared__ double acc[T * 64];
    __shared__ double red[T];
    __shared__ double wl[T];
    const int mode = flags & GABO_OUT_MASK;
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x / n1;
    const int64_t i = blockIdx.x - b * n1;
    const double* W = Winv + b * w_batch_stride + i * T;
    static_for<T>([&](auto ee) { acc[decltype(ee)::value * 64 + lane] = 0.0; });
    for (int64_t j0 = 0; j0 < n2; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool live = j < n2;
        const int64_t jc = live ? j : n2 - 1;
        const double* Gj = G + b * g_batch_stride + jc;
        // M = C C^T, C = W G_j (same construction as the forward kernel)
        double m[T];
        static_for<T>([&](auto ee) { m[decltype(ee)::value] = 0.0; });
        static_for<D>([&](auto cc) {
            constexpr int col = decltype(cc)::value;
            double g[D - col], c[D - col];
            static_for<D - col>([&](auto kk) { g[decltype(kk)::value] = Gj[(int64_t)tri(col + decltype(kk)::value, col) * n2]; });
            static_for<D - col>([&](auto rr) {
                constexpr int r = col + decltype(rr)::value;
                double a = W[tri(r, col)] * g[0];
                static_for<r - col>([&](auto kk) {
                    constexpr int k = col + 1 + decltype(kk)::value;
                    a = __builtin_fma(W[tri(r, k)], g[k - col], a);
                });
                c[r - col] = a;
            });
            static_for<D - col>([&](auto rr) {
                constexpr int r = col + decltype(rr)::value;
                static_for<r - col + 1>([&](auto qq) {
                    constexpr int q = col + decltype(qq)::value;
                    m[tri(r, q)] = __builtin_fma(c[r - col], c[q - col], m[tri(r, q)]);
                });
            });
        });
        // eigen-decomposition in registers: Householder + QL with vectors (spd_eigvec.hpp)
        constexpr bool kJacobi = D <= GABO_BWD_JACOBI_MAX_DIM;
        double vreg[D * D];
        double lam[D];
        if constexpr (kJacobi) {
            jacobi_eig_reg<D>(m, vreg);
            static_for<D>([&](auto kk) { lam[decltype(kk)::value] = m[tri(decltype(kk)::value, decltype(kk)::value)]; });
        } else {
            // (threshold |e| <= 1e-13 (|d| + |d'|) instead of machine epsilon: logm(M) moves by e f[l_k, l_k+1] <= 1e-13 |log'|, three orders below what a
            // gradient is compared at - and the last sweep of some stages is saved: N = 4096, d = 10 same-box 9.24 -> 9.00 ms, max error / max |grad| against the
            // oracle 3e-15 -> 5e-14 (1e-22: 8.78 ms, 3e-12; tools/ab_bwd_eps_r05.sh); the trust-region kernels keep the strict form, their traces are pinned)
#ifndef GABO_BWD_QL_EPS2
#define GABO_BWD_QL_EPS2 1e-26
#endif
            double qe = GABO_BWD_QL_EPS2;
            asm volatile("" : "+s"(qe));
            // (round 6: eigenvalues first, then ONE vector-accumulating sweep per stage shifted by its known eigenvalue - spd_eigvec.hpp,
            // sym_eig_reg_two_pass; -DGABO_BWD_TWO_PASS_MIN_DIM=99 keeps the one-pass solver)
#ifndef GABO_BWD_TWO_PASS_MIN_DIM
#define GABO_BWD_TWO_PASS_MIN_DIM 4
#endif
            if constexpr (D >= GABO_BWD_TWO_PASS_MIN_DIM) sym_eig_reg_two_pass<D>(m, lam, vreg, qe);
            else sym_eig_reg<D>(m, lam, vreg, qe);
        }
        auto Vat = [&](int r, int c) -> double { return vreg[r * D + c]; };
        double lg[D];
        double s = 0.0;
#ifdef GABO_BWD_OCML_LOG       /* A/B: OCML's log (98 instructions, double-double) as in rounds 1-4 */
        static_for<D>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            lg[k] = log(lam[k]);
            s = __builtin_fma(lg[k], lg[k], s);
        });
#else
        {
            // the fdlibm-scheme log of the acquisition kernels (gabo_device.hpp: ~35 instructions at 1 ulp) instead of OCML's 98: D of them per pair.
            // An eigenvalue that is not positive (M = C C^T is positive semi-definite by construction: rounding of a singular pair, or NaN factors of
            // a matrix of x2 that is not positive definite) gives NaN, as torch.log does in spd_utils_torch.py:115.
            const LogRegs logc = LogRegs::load();
            static_for<D>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                const double l = log_pos(lam[k], logc);
                lg[k] = lam[k] > 0.0 ? l : __builtin_nan("");
                s = __builtin_fma(lg[k], lg[k], s);
            });
        }
#endif
        // w = dLoss/d(d^2)
        double d2 = s + 1e-15;
        double go = live ? gout[b * go_sb + i * go_si + j * go_sj] : 0.0;
        double w;
        if (mode == GABO_OUT_GAUSSIAN) {
            double dist = __builtin_sqrt(d2);
            w = go * (-beta) * exp(-((dist * dist) * beta));
        } else if (mode == GABO_OUT_LAPLACE) {
            double dist = __builtin_sqrt(d2);
            w = go * (-beta) * exp(-(dist * beta)) / (2.0 * dist);
        } else {
            w = go / (2.0 * __builtin_sqrt(d2));
        }
        // acc += w * V diag(lg) V^T   (lower triangle)
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            double vl[D];                                   // row r of V diag(w lg)
            static_for<D>([&](auto kk) { vl[decltype(kk)::value] = Vat(r, decltype(kk)::value) * (w * lg[decltype(kk)::value]); });
            static_for<r + 1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                double f = acc[tri(r, c) * 64 + lane];
                static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; f = __builtin_fma(vl[k], Vat(c, k), f); });
                acc[tri(r, c) * 64 + lane] = f;
            });
        });
    }
    // publish the sums and stage W in LDS for the dynamic-index congruence
    __syncthreads();
    for (int e = lane; e < T; e += 64) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += acc[e * 64 + ((l + e) & 63)];  // rotated start: threads hit different banks
        red[e] = t;
    }
    for (int e = lane; e < T; e += 64) wl[e] = W[e];
    __syncthreads();
    // grad_A = -2 W^T S W (symmetric); thread e owns entry (a, bb), a >= bb.  W lower: W[r][a] != 0 only for r >= a.
    for (int e = lane; e < T; e += 64) {
        int a = 0;
        while (tri(a + 1, 0) <= e) ++a;
        int bb = e - tri(a, 0);
        double t = 0.0;
        for (int r = a; r < D; ++r) {
            double inner = 0.0;
            for (int c = bb; c < D; ++c) {
                double srs = r >= c ? red[tri(r, c)] : red[tri(c, r)];
                inner = __builtin_fma(srs, wl[tri(c, bb)], inner);
            }
            t = __builtin_fma(wl[tri(r, a)], inner, t);
        }
        t *= -2.0;
        // Mandel: diagonal as is, off-diagonal * sqrt(2)  (= 0.5 (sqrt2 G_rc + sqrt2 G_cr), spd_utils_torch.py:219)
        gx[(b * n1 + i) * T + mandel_pos(D, a, bb)] = (a == bb) ? t : t * kSqrt2;
    }
}

template <int D>
static int launch_spd_ai_backward(const double* x1, const double* x2, const double* gout, double* gx, int64_t batch, int64_t n1,
                                  int64_t n2, int64_t s1, int64_t s2, int64_t go_sb, int64_t go_si, int64_t go_sj, double beta,
                                  int flags, double* ws, int* status, hipStream_t st) {
    constexpr int T = tri_size(D);
    const int64_t b1 = (s1 == 0) ? 1 : batch;
    const int64_t b2 = (s2 == 0) ? 1 : batch;
    double* W = ws;
    double* G = ws + b1 * n1 * T;
    launch_spd_prep<D>(x1, x2, W, G, b1, b2, n1, n2, s1, s2, status, st);
    int64_t nblocks = batch * n1;
    if (nblocks > 0x7fffffffLL) return GABO_ERR_ARG;
    hipLaunchKernelGGL((spd_ai_backward_kernel<D>), dim3((unsigned)nblocks), dim3(64), 0, st, W, G, gout, gx, n1, n2,
                       (s1 == 0) ? (int64_t)0 : n1 * T, (s2 == 0) ? (int64_t)0 : n2 * T, go_sb, go_si, go_sj, beta, flags);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

// d = 12 ... 16: two lanes per pair (spd_backward_duo.hip)
int launch_spd_ai_backward_duo(int d, const double* x1, const double* x2, const double* gout, double* gx, int64_t batch, int64_t n1, int64_t n2,
                               int64_t s1, int64_t s2, int64_t go_sb, int64_t go_si, int64_t go_sj, double beta, int flags, double* ws, int* status,
                               hipStream_t st);

}  // namespace gabo

#ifndef GABO_BWD_DUO_MIN_DIM
#define GABO_BWD_DUO_MIN_DIM 12    /* two lanes per pair from here (d = 12: 22.9 against 24.3 ms; below, the one-lane kernel is faster: spd_backward_duo.hip) */
#endif

extern "C" int gabo_spd_ai_backward(const double* x1, const double* x2, const double* grad_out, double* grad_x1, int64_t batch,
                                    int64_t n1, int64_t n2, int d, int64_t x1_batch_stride, int64_t x2_batch_stride,
                                    int64_t go_batch_stride, int64_t go_row_stride, int64_t go_col_stride, double beta, int flags,
                                    void* workspace, size_t workspace_bytes, int* status, gabo_stream_t stream) {
    if (batch < 0 || n1 < 0 || n2 < 0 || x1_batch_stride < 0 || x2_batch_stride < 0) return GABO_ERR_ARG;
    if (d < 2 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (batch == 0 || n1 == 0) return GABO_OK;
    if (!x1 || !grad_x1 || !workspace || !status) return GABO_ERR_ARG;
    if (n2 > 0 && (!x2 || !grad_out)) return GABO_ERR_ARG;
    if (workspace_bytes < gabo_spd_ai_workspace_bytes(batch, n1, n2, d)) return GABO_ERR_ARG;
    if (flags & GABO_SYMMETRIC) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
#ifndef GABO_ONLY_DIM
    if (d >= GABO_BWD_DUO_MIN_DIM && d <= GABO_SPD_BWD_REG_MAX_DIM)
        return gabo::launch_spd_ai_backward_duo(d, x1, x2, grad_out, grad_x1, batch, n1, n2, x1_batch_stride, x2_batch_stride, go_batch_stride,
                                                go_row_stride, go_col_stride, beta, flags, ws, status, st);
#endif
    if (d > GABO_SPD_REG_MAX_DIM)
        return gabo::launch_spd_ai_backward_generic(x1, x2, grad_out, grad_x1, batch, n1, n2, d, x1_batch_stride, x2_batch_stride,
                                                    go_batch_stride, go_row_stride, go_col_stride, beta, flags, ws, status, st);
#define GABO_CASE(DD) \
    case DD:          \
        return gabo::launch_spd_ai_backward<DD>(x1, x2, grad_out, grad_x1, batch, n1, n2, x1_batch_stride, x2_batch_stride, \
                                                go_batch_stride, go_row_stride, go_col_stride, beta, flags, ws, status, st);
    switch (d) {
#ifdef GABO_ONLY_DIM
        GABO_CASE(GABO_ONLY_DIM)
#else
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
#endif
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}
