// Closed-form gradient of the SPD affine-invariant pairwise kernels (see spd_backward.hip) for d = 12 ... 16 with TWO lanes per pair.
//
// The one-lane kernel keeps the eigenvector matrix Z of every pair in its lane's registers: D^2 doubles - 288 VGPRs at d = 12 - and stops at
// d = 12 (512 registers per lane).  Here lanes 2p and 2p + 1 share pair p of the wave's 32 columns; each holds HALF the rows of Z (spd_eigvec.hpp:
// sym_eig_duo - the QL rotations, two thirds of the work, act on columns and need no exchange; the reflector accumulation exchanges one partial dot
// product per column through a DPP swap; congruence, tridiagonalisation and the scalar QL recurrence run redundantly and bit-identically in both
// lanes).  Half of Z fits 512 registers up to d = 16, where the path used to fall to one WAVE per pair with LDS tiles (d = 13: 1.8e7 -> 6.9e8 pairs/s).
// What it does NOT do is speed up d <= 11: measured (tools/ab_backward.py, PMC in profiles/r03_pmc_backward.txt) the one-lane kernel at ONE wave per
// SIMD already issues at 4.6 cycles per wave-instruction - its rotations are independent across the rows of Z, and a wave with that much
// instruction-level parallelism is not limited to every other issue slot as a dependent chain is -, so a second wave per SIMD has nothing to fill
// (duo at d = 10: 1.43x the wave-instructions, 12.2 ms with two waves per SIMD, 13.4 with one, against 8.7 for the one-lane kernel).
// logm(M) = V diag(log lambda) V^T is accumulated entry by entry: entries (r, c) with r, c of the lane's own parity from its own rows, the
// mixed-parity ones from its rows and the partner's (fetched row by row through the same swap); the two lanes of a pair own disjoint entries of one shared LDS
// column, so the reduction over the wave and the final congruence are as before.
#include "gabo_device.hpp"
#include "spd_prep.hpp"
#include "spd_eigvec.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

#ifndef GABO_DUO_WAVES
#define GABO_DUO_WAVES 2
#endif
template <int D>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((D > 12 ? 1 : GABO_DUO_WAVES), (D > 12 ? 1 : GABO_DUO_WAVES)))) void spd_ai_backward_duo_kernel(const double* __restrict__ Winv, const double* __restrict__ G,
                                                                                  const double* __restrict__ gout, double* __restrict__ gx, int64_t n1,
                                                                                  int64_t n2, int64_t w_batch_stride, int64_t g_batch_stride,
                                                                                  int64_t go_sb, int64_t go_si, int64_t go_sj, double beta, int flags) {
    constexpr int T = tri_size(D);
    constexpr int HR = DuoShape<D>::HR;
    constexpr int HRO = D / 2;                       // odd rows 1, 3, ...: HRO of them (HR even rows)
    // one column per lane PAIR: the two lanes own disjoint entries of w logm(M) (d = 10: 14 KB per wave - eight waves per CU fit; d = 16: 35 KB, four)
    __shared__ double acc[T * 32];
    __shared__ double red[T];
    __shared__ double wl[T];
    const int mode = flags & GABO_OUT_MASK;
    const int lane = threadIdx.x;
    const bool h = lane & 1;
    const int64_t b = blockIdx.x / n1;
    const int64_t i = blockIdx.x - b * n1;
    const double* W = Winv + b * w_batch_stride + i * T;
    const int col = lane >> 1;
    for (int e = lane; e < T * 32; e += 64) acc[e] = 0.0;
    __syncthreads();
    for (int64_t j0 = 0; j0 < n2; j0 += 32) {
        const int64_t j = j0 + (lane >> 1);
        const bool live = j < n2;
        const int64_t jc = live ? j : n2 - 1;
        const double* Gj = G + b * g_batch_stride + jc;
        // M = C C^T, C = W G_j (the forward kernel's construction; both lanes of the pair)
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
        double vh[HR * D];
        double lam[D];
        sym_eig_duo<D>(m, lam, vh, h);
        // logarithms: each lane takes every other eigenvalue, the partner's come through the swap
        double lg[D];
        const LogRegs logc = LogRegs::load();
        static_for<HR>([&](auto kk) {
            constexpr int ke = 2 * decltype(kk)::value, ko = ke + 1 < D ? ke + 1 : ke;
            const double lsel = h ? lam[ko] : lam[ke];
            const double mine = lsel > 0.0 ? log_pos(lsel, logc) : __builtin_nan("");   // (as spd_backward.hip: 35 instructions instead of OCML's 98)
            const double other = duo_swap(mine);
            lg[ke] = h ? other : mine;
            if constexpr (ke + 1 < D) lg[ke + 1] = h ? mine : other;
        });
        double s = 0.0;
        static_for<D>([&](auto kk) { s = __builtin_fma(lg[decltype(kk)::value], lg[decltype(kk)::value], s); });
        // w = dLoss/d(d^2)
        const double d2 = s + 1e-15;
        const double go = live ? gout[b * go_sb + i * go_si + j * go_sj] : 0.0;
        double w;
        if (mode == GABO_OUT_GAUSSIAN) {
            const double dist = __builtin_sqrt(d2);
            w = go * (-beta) * exp(-((dist * dist) * beta));
        } else if (mode == GABO_OUT_LAPLACE) {
            const double dist = __builtin_sqrt(d2);
            w = go * (-beta) * exp(-(dist * beta)) / (2.0 * dist);
        } else {
            w = go / (2.0 * __builtin_sqrt(d2));
        }
        double wlg[D];
        static_for<D>([&](auto kk) { wlg[decltype(kk)::value] = w * lg[decltype(kk)::value]; });
        // (i) entries whose rows both have this lane's parity: (2a + h, 2b + h), b <= a.  For odd D the odd lane's last local row is padding:
        //     that lane skips the update
        static_for<HR>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            double pa[D];
            static_for<D>([&](auto kk) { pa[decltype(kk)::value] = vh[a * D + decltype(kk)::value] * wlg[decltype(kk)::value]; });
            static_for<a + 1>([&](auto bb) {
                constexpr int bq = decltype(bb)::value;
                constexpr int ie = tri(2 * a, 2 * bq);
                constexpr bool odd_ok = 2 * a + 1 < D;
                constexpr int io = odd_ok ? tri(2 * a + 1, 2 * bq + 1) : ie;
                const int idx = (h ? io : ie) * 32 + col;
                double f = 0.0;
                static_for<D>([&](auto kk) { f = __builtin_fma(pa[decltype(kk)::value], vh[bq * D + decltype(kk)::value], f); });
                if (odd_ok || !h) acc[idx] += f;
            });
        });
        // (ii) mixed parity: this lane's local row a with the partner's local row bq >= a.  In even-lane coordinates the even lane covers the
        //      (even 2a, odd 2bq + 1) entries with a <= bq and the odd lane the (even 2bq, odd 2a + 1) ones: together every mixed entry once, the
        //      a = bq ones twice - there the odd lane's weight is zero
        static_for<HR>([&](auto bb) {
            constexpr int bq = decltype(bb)::value;
            double qb[D];
            static_for<D>([&](auto kk) { qb[decltype(kk)::value] = duo_swap(vh[bq * D + decltype(kk)::value]) * wlg[decltype(kk)::value]; });
            static_for<bq + 1>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                // even lane: rows (2a, 2bq+1) -> tri(2bq+1, 2a) if that odd row exists; odd lane: rows (2a+1, 2bq) -> a == bq: tri(2a+1, 2a), a < bq: tri(2bq, 2a+1)
                constexpr bool even_ok = 2 * bq + 1 < D, odd_ok = 2 * a + 1 < D;
                constexpr int ie = even_ok ? tri(2 * bq + 1, 2 * a) : 0;
                constexpr int io = odd_ok ? (a == bq ? tri(2 * a + 1, 2 * a) : tri(2 * bq, 2 * a + 1)) : 0;
                if constexpr (even_ok || odd_ok) {
                    const int idx = (h ? io : ie) * 32 + col;
                    double f = 0.0;
                    static_for<D>([&](auto kk) { f = __builtin_fma(vh[a * D + decltype(kk)::value], qb[decltype(kk)::value], f); });
                    bool keep = h ? odd_ok : even_ok;
                    if constexpr (a == bq) keep = keep && !h;
                    if (keep) acc[idx] += f;
                }
            });
        });
    }
    // publish the sums and stage W in LDS for the dynamic-index congruence
    __syncthreads();
    for (int e = lane; e < T; e += 64) {
        double t = 0.0;
        for (int l = 0; l < 32; ++l) t += acc[e * 32 + ((l + e) & 31)];  // rotated start: threads hit different banks
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
        gx[(b * n1 + i) * T + mandel_pos(D, a, bb)] = (a == bb) ? t : t * kSqrt2;
    }
}

template <int D>
static int launch_duo(const double* x1, const double* x2, const double* gout, double* gx, int64_t batch, int64_t n1, int64_t n2, int64_t s1,
                      int64_t s2, int64_t go_sb, int64_t go_si, int64_t go_sj, double beta, int flags, double* ws, int* status, hipStream_t st) {
    constexpr int T = tri_size(D);
    const int64_t b1 = (s1 == 0) ? 1 : batch;
    const int64_t b2 = (s2 == 0) ? 1 : batch;
    double* W = ws;
    double* G = ws + b1 * n1 * T;
    launch_spd_prep<D>(x1, x2, W, G, b1, b2, n1, n2, s1, s2, status, st);
    int64_t nblocks = batch * n1;
    if (nblocks > 0x7fffffffLL) return GABO_ERR_ARG;
    hipLaunchKernelGGL((spd_ai_backward_duo_kernel<D>), dim3((unsigned)nblocks), dim3(64), 0, st, W, G, gout, gx, n1, n2,
                       (s1 == 0) ? (int64_t)0 : n1 * T, (s2 == 0) ? (int64_t)0 : n2 * T, go_sb, go_si, go_sj, beta, flags);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int launch_spd_ai_backward_duo(int d, const double* x1, const double* x2, const double* gout, double* gx, int64_t batch, int64_t n1, int64_t n2,
                               int64_t s1, int64_t s2, int64_t go_sb, int64_t go_si, int64_t go_sj, double beta, int flags, double* ws, int* status,
                               hipStream_t st) {
#define GABO_CASE(DD) \
    case DD:          \
        return launch_duo<DD>(x1, x2, gout, gx, batch, n1, n2, s1, s2, go_sb, go_si, go_sj, beta, flags, ws, status, st);
    switch (d) {
#ifdef GABO_ONLY_DIM
        GABO_CASE(GABO_ONLY_DIM)
#else
        GABO_CASE(12) GABO_CASE(13) GABO_CASE(14) GABO_CASE(15) GABO_CASE(16)
#endif
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

}  // namespace gabo
