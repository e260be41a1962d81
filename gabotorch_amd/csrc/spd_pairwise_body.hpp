// Kernel templates of the SPD pairwise Gram (see spd_pairwise.hip for the description).
#pragma once
#include "gabo_device.hpp"
#include "spd_eig.hpp"
#include "spd_prep.hpp"
#include "gabo_mirror.hpp"
#include "spd_generic.hpp"
#include "gabo_log_tab.hpp"
#include "gabo_exp_tab256.hpp"
#include "gabo_acosh2_table.hpp"
#include "../../include/gabo_hip.h"

#ifndef GABO_PAIR_WAVES
#define GABO_PAIR_WAVES 2  /* waves per SIMD the pairwise kernel is register-budgeted for */
#endif

namespace gabo {

// The lane's column of G as a pointer into the GLOBAL address space.  The row loop launders this pointer (so that the G loads are not
// hoisted); laundering a plain `const double*` makes it a generic pointer, the loads become flat_load - which count in vmcnt AND lgkmcnt and
// may return out of order - and every wait in front of their uses becomes `s_waitcnt vmcnt(0) lgkmcnt(0)`, a full drain four times per row.
// With the address space kept the loads are global_load and the waits are the partial vmcnt(N) of in-order returns.
#ifdef GABO_PAIR_FLAT   /* A/B: the round-2 form */
typedef const double* spd_gcol_ptr;
#else
typedef const double __attribute__((address_space(1)))* spd_gcol_ptr;
#endif

// sum_k log^2(lambda_k) of M = C C^T with C = W * G (both lower triangular, W wave-uniform, G per lane)
// The same column through a buffer descriptor: address = wave-uniform base (in the descriptor) + the lane's byte offset + a SCALAR offset per
// entry.  The 55 entry offsets k * n2 * 8 are then computed on the scalar unit; with a vector pointer each load needs its own 64-bit vector
// add (v_lshl_add_u64), 54 of them per pair at d = 10.  Usable while T * n2 * 8 < 2^31.
struct SpdGcolBuffer {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;            // laundered once per row (keeps the loads inside the row loop)
    int stride_bytes;    // n2 * 8
    __device__ __forceinline__ double operator[](int k) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, k * stride_bytes, 0));
    }
};
struct SpdGcolPointer {
    spd_gcol_ptr p;
    int64_t stride;
    __device__ __forceinline__ double operator[](int k) const { return p[(int64_t)k * stride]; }
};

template <int D, class GCOL>
__device__ __forceinline__ double ai_sumsq(const double* __restrict__ W, const GCOL& Gj, const double* __restrict__ ltab, const double eps2 = 0.0) {
    constexpr int T = tri_size(D);
    // Column `col` of C = W G depends only on column `col` of G:  C[r][col] = sum_{k=col..r} W[r][k] G[k][col].
    // M = C C^T = sum_col C[:,col] C[:,col]^T, so M is accumulated by rank-1 updates and C is never held whole:
    // live state is M (T doubles) + one column of C + one column of G.
    double m[T];
    static_for<D>([&](auto cc) {
        constexpr int col = decltype(cc)::value;
        double g[D - col], c[D - col];
        static_for<D - col>([&](auto kk) { g[decltype(kk)::value] = Gj[tri(col + decltype(kk)::value, col)]; });
        static_for<D - col>([&](auto rr) {
            constexpr int r = col + decltype(rr)::value;
            double acc = W[tri(r, col)] * g[0];
            static_for<r - col>([&](auto kk) {
                constexpr int k = col + 1 + decltype(kk)::value;
                acc = __builtin_fma(W[tri(r, k)], g[k - col], acc);
            });
            c[r - col] = acc;
        });
        static_for<D - col>([&](auto rr) {
            constexpr int r = col + decltype(rr)::value;
            static_for<r - col + 1>([&](auto qq) {
                constexpr int q = col + decltype(qq)::value;
                // column 0 touches every entry of M first: it initialises, the later columns accumulate
                m[tri(r, q)] = (col == 0) ? c[r - col] * c[q - col] : __builtin_fma(c[r - col], c[q - col], m[tri(r, q)]);
            });
        });
    });
    double dg[D], e2[D];
    tridiagonalize<D>(m, dg, e2);
    tridiag_eigenvalues<D>(dg, e2, eps2);
    double s = 0.0;
#ifdef GABO_OCML_LOG
    static_for<D>([&](auto kk) { double lg = log(dg[decltype(kk)::value]); s = __builtin_fma(lg, lg, s); });
#elif defined(GABO_LOG_FDLIBM)                /* A/B: the round-1 log (fdlibm scheme, 35 instructions with a v_rcp_f64) */
    const LogRegs lr = LogRegs::load();       // pinned here, after M and the tridiagonal are dead: no extra register pressure
    static_for<D>([&](auto kk) { double lg = log_pos(dg[decltype(kk)::value], lr); s = __builtin_fma(lg, lg, s); });
#else
    const LogTabRegs lr = LogTabRegs::load();
    static_for<D>([&](auto kk) { double lg = log_tab(dg[decltype(kk)::value], lr, ltab); s = __builtin_fma(lg, lg, s); });
#endif
    return s;
}

// d = 2 (the latent space of the nested kernels, config 5) in closed form with ONE logarithm per pair: M = C C^T, C = W G lower triangular, so
// det M = (c00 c11)^2 exactly and log lambda_- = log det M - log lambda_+; log det M = 2 (log(w00 w11) + log(g00 g11)) is a per-POINT quantity
// (`lw2`: the row's, wave-uniform; `lg2`: the lane's column's, computed once before the row loop).  lambda_+ = (tr + sqrt((m00 - m11)^2 + 4 m10^2)) / 2
// has no cancellation.  ~45 instructions per pair against ~120 for the generic path (two logs, the 2x2 QL finish, three operand loads).
struct Spd2Col {
    double g00, g10, g11, lg;
};
__device__ __forceinline__ double ai_sumsq2(const double* __restrict__ W, const Spd2Col& g, double lw, const LogTabRegs& lr,
                                            const double* __restrict__ ltab, double tiny) {
    const double c00 = W[0] * g.g00, c11 = W[2] * g.g11;
    const double c10 = __builtin_fma(W[1], g.g00, W[2] * g.g10);
    const double m00 = c00 * c00, m10d = (c10 + c10) * c00, m11 = __builtin_fma(c10, c10, c11 * c11);
    const double tr = m00 + m11, df = m00 - m11;
    // (`tiny` = 1e-290 keeps the seed-based square root away from 0 when M is a multiple of the identity)
    const double root = sqrt_nz(__builtin_fma(df, df, __builtin_fma(m10d, m10d, tiny)));
    const double l1 = log_tab(0.5 * (tr + root), lr, ltab);
    const double l2 = __builtin_fma(2.0, lw + g.lg, -l1);
    return __builtin_fma(l1, l1, l2 * l2);
}

__device__ __forceinline__ double finish(double dist, double beta, int mode) {
    if (mode == GABO_OUT_DISTANCE) return dist;
    if (mode == GABO_OUT_LAPLACE) return exp(-(dist * beta));   // kernels_spd.py:185
    return exp(-((dist * dist) * beta));                        // kernels_spd.py:94-98
}

// 1-D grid, block id -> (batch, row chunk of `rows` rows, column group of blockDim.x columns), column group fastest
#ifndef GABO_PAIR_TWO_WAVE_MAX_DIM
// Two waves per SIMD (256 VGPRs) up to d = 14: at d = 13 / 14 the compiler spills 10 / 24 doubles per lane to scratch, which costs far less than
// the halved issue rate of a lone wave (8.5 instead of 4.5 cycles per instruction): N = 4096 Gram 5.92 -> 4.33 ms and 7.77 -> 5.87 ms; d = 15: 9.9 ->
// 9.8, d = 16: 11.9 -> 14.1 (one wave per SIMD from 15 on).
#define GABO_PAIR_TWO_WAVE_MAX_DIM 14
#endif
template <int D>
__global__ __launch_bounds__(256, (D > GABO_PAIR_TWO_WAVE_MAX_DIM ? 1 : GABO_PAIR_WAVES)) void spd_ai_pairwise_kernel(const double* __restrict__ Winv, const double* __restrict__ G,
                                                              double* __restrict__ out, double* __restrict__ dist_out,
                                                              int64_t n1, int64_t n2,
                                                              int64_t w_batch_stride, int64_t g_batch_stride, int rows,
                                                              int col_blocks, int row_chunks, int64_t sym_tiles, double beta, int flags) {
    constexpr int T = tri_size(D);
    const int mode = flags & GABO_OUT_MASK;
    // small matrices (the latent spaces of the nested kernels: d = 2, 3): the finish is a visible part of a pair's ~140 instructions,
    // so the Gaussian mode uses the table-assisted exp of the write-bound kernels (17 instead of ~37 instructions).  From d = 5 on the
    // scalar registers are taken by the W row and OCML's exp measured faster than any variant with pinned coefficients.
    constexpr bool kTabExp = D <= 4;
    __shared__ double tab[kTabExp ? 64 : 1];
    __shared__ __attribute__((aligned(16))) double ltab[512];        // (c, -log c) pairs of log_tab
    for (int k = threadIdx.x; k < 512; k += blockDim.x) ltab[k] = kLogTab[k];
    double ec[6], ec3 = 0.0;
    if constexpr (kTabExp) {
        if (threadIdx.x < 64) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
    }
    __syncthreads();
    if constexpr (kTabExp) {
        static_for<6>([&](auto k) { ec[decltype(k)::value] = kExpTabC[decltype(k)::value]; });
        ec3 = ec[3];
        asm volatile("" : "+v"(ec3));
    }
    int64_t cg, rc, b;
    if (flags & GABO_SYMMETRIC) {
        // Only tiles touching the upper triangle exist in the grid: column group cg owns row chunks
        // [0, min(row_chunks, ceil((cg+1)*cols/rows))).  Enumerating exactly those keeps consecutive block ids
        // (= consecutive XCDs) equally loaded; skipping blocks of a full grid instead leaves XCD 0 with 1/3 of
        // the work of XCD 7.
        const int64_t per_batch = sym_tiles;
        b = blockIdx.x / per_batch;
        int64_t t = blockIdx.x - b * per_batch;
        cg = 0;
        for (;;) {
            int64_t cnt = ((cg + 1) * (int64_t)blockDim.x + rows - 1) / rows;
            if (cnt > row_chunks) cnt = row_chunks;
            if (t < cnt) break;
            t -= cnt;
            ++cg;
        }
        rc = t;
    } else {
        const int64_t bid = blockIdx.x;
        cg = bid % col_blocks;
        rc = (bid / col_blocks) % row_chunks;
        b = bid / ((int64_t)col_blocks * row_chunks);
    }
    const int64_t j0 = cg * blockDim.x;
    const int64_t j = j0 + threadIdx.x;
    const int64_t jc = j < n2 ? j : n2 - 1;  // out-of-range lanes recompute the last column and do not store
    const int64_t i0 = rc * rows;
    const int64_t i1 = (i0 + rows < n1) ? i0 + rows : n1;
    // symmetric mode, tile straddling the diagonal: a wave whose 64 columns all lie left of the tile's first row has nothing
    // to store (i > j for every pair it would evaluate)
    if ((flags & GABO_SYMMETRIC) && j0 + (int64_t)(threadIdx.x | 63) < i0) return;
    const double* Gj = G + b * g_batch_stride + jc;
    double* ob = out + b * n1 * n2;
    // buffer form of the column loads (see SpdGcolBuffer): base = the wave's first column, lane offset = its (clamped) column
    const int64_t jw = j0 + (threadIdx.x & ~63);
    const bool use_buffer = D > 2 && (int64_t)T * n2 * 8 < (1ll << 31);
    SpdGcolBuffer gbuf;
    {
        const double* gw = G + b * g_batch_stride + (jw < n2 ? jw : n2 - 1);
        uint64_t a = (uint64_t)gw;
        a = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
        gbuf.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x7fffffff, 0x00020000);
        gbuf.voff = (int)((jc - (jw < n2 ? jw : n2 - 1)) * 8);
        gbuf.stride_bytes = (int)(n2 * 8);
    }
    // d = 2: the lane's column (3 numbers) and its log-determinant term stay in registers; lane r of every wave holds the term of row i0 + r
    Spd2Col col2;
    double lw_lane = 0.0, tiny = 1e-290;
    LogTabRegs lr2;
    if constexpr (D == 2) {
        lr2 = LogTabRegs::load();
        asm volatile("" : "+s"(tiny));
        col2.g00 = Gj[0];
        col2.g10 = Gj[n2];
        col2.g11 = Gj[2 * n2];
        col2.lg = log_tab(col2.g00 * col2.g11, lr2, ltab);
        int64_t ir = i0 + (threadIdx.x & 63);
        ir = ir < n1 ? ir : n1 - 1;
        const double* Wr = Winv + b * w_batch_stride + ir * T;
        lw_lane = log_tab(Wr[0] * Wr[2], lr2, ltab);
    }
    // deflation threshold of the QL iteration: 1e-20 where a distance leaves the kernel (distance and Laplace modes, or a distance output next to the
    // Gaussian values); GABO_QL_EPS2_GAUSS for Gaussian values alone (see tridiag_eigenvalues)
#ifndef GABO_QL_EPS2_GAUSS
#define GABO_QL_EPS2_GAUSS 1e-15
#endif
    double eps2 = (mode == GABO_OUT_GAUSSIAN && !dist_out) ? GABO_QL_EPS2_GAUSS : 0.0;
    asm volatile("" : "+s"(eps2));
    for (int64_t i = i0; i < i1; ++i) {
        const double* W = Winv + b * w_batch_stride + i * T;
        double s;
        if constexpr (D == 2) {
            const int lane = (int)(i - i0);
            const double lw = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lw_lane), lane),
                                               __builtin_amdgcn_readlane(__double2loint(lw_lane), lane));
            s = ai_sumsq2(W, col2, lw, lr2, ltab, tiny);
        } else {
            // Launder the column pointer so the 55 G loads are NOT hoisted out of the row loop: keeping G resident costs
            // 110 VGPRs (one wave per SIMD less); re-reading it from L2 costs 28 KB per wave-row, which is noise here.
#ifndef GABO_PAIR_NO_BUFFER
            if (use_buffer) {
                SpdGcolBuffer gb = gbuf;
                asm volatile("" : "+v"(gb.voff));
                asm volatile("" : "+s"(gb.stride_bytes));      // the entry offsets are recomputed per row on the scalar unit, not kept (the W row owns the SGPRs)
                s = ai_sumsq<D>(W, gb, ltab, eps2);
            } else
#endif
            {
                SpdGcolPointer gp{(spd_gcol_ptr)Gj, n2};
                asm volatile("" : "+v"(gp.p));
                s = ai_sumsq<D>(W, gp, ltab, eps2);
            }
        }
        double dist, val;
        if (mode == GABO_OUT_GAUSSIAN && !dist_out) {
            // exp(-beta sqrt(s + 1e-15)^2) (spd_utils_torch.py:120, kernels_spd.py:94-98) without the square root: sqrt(x)^2 = x to an ulp,
            // i.e. 1e-16 beta d^2 relative in K.  (OCML's exp stays: the register-table exp_neg of gabo_device.hpp measured 4 % SLOWER here,
            // 2.75 vs 2.63 ms - its 28 pinned coefficients cost more than the instructions they save once per pair.)
            dist = 0.0;
            if constexpr (kTabExp) {
                val = exp_neg_tab(-((s + 1e-15) * beta), ec, ec3, tab);
                val = s != s ? s : val;        // (the argument clamp of the table exp returns its bound for a NaN: a NaN column of x2 must stay NaN)
            } else {
                val = exp(-((s + 1e-15) * beta));
            }
        } else {
            dist = __builtin_sqrt(s + 1e-15);  // spd_utils_torch.py:120
            val = finish(dist, beta, mode);
        }
        // symmetric mode: only the upper triangle (i <= j) is stored; mirror_upper_kernel fills the rest afterwards,
        // so the result is exactly symmetric and the mirror writes are coalesced.
        if (j < n2 && (!(flags & GABO_SYMMETRIC) || i <= j)) {
            ob[i * n2 + j] = val;
            if (dist_out) dist_out[b * n1 * n2 + i * n2 + j] = dist;
        }
    }
}

// ---- d = 2, Gaussian kernel values only (no distance output): the Gram of the nested kernels' latent space (config 5) ----------------------
// The same closed form as ai_sumsq2, in a kernel of its own so that the row loop contains nothing else: lane r of every wave holds row i0 + r
// of W and its log-determinant term (read back with v_readlane: no scalar-memory round trip per row), the column in registers, log of
// (tr + root) with the halving folded into the exponent, the cubic square root, exp with magic-number rounding on the 256-entry table
// (sphere_pairwise.hip), beta folded into one FMA, streaming stores for results beyond L2.
constexpr double kMagicRound = 6755399441055744.0;      // 1.5 2^52
// [0..3] ln2/256 head and tail, -256/ln2, 1/6 (exp_of_minus_tab256_magic)
__constant__ double kExpC256[4] = {0.010830424695086549 / 4, 1.162596423439437e-12 / 4, -92.33248261689366 * 4, 1.0 / 6.0};

// (round 2: the sqrt + log form, kept for A/B builds: -DGABO_GAUSS2_LOG)
template <bool NT>
__global__ __launch_bounds__(256) void spd_ai_gauss2_log_kernel(const double* __restrict__ Winv, const double* __restrict__ G,
                                                            double* __restrict__ out, int64_t n1, int64_t n2, int64_t w_batch_stride,
                                                            int64_t g_batch_stride, int rows, int col_blocks, int row_chunks,
                                                            int64_t sym_tiles, double beta, int flags) {
    __shared__ __attribute__((aligned(16))) double ltab[512];
    __shared__ double etab[256];
    for (int k = threadIdx.x; k < 512; k += blockDim.x) ltab[k] = kLogTab[k];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) etab[k] = kExp2Tab256[k];
    int64_t cg, rc, b;
    if (flags & GABO_SYMMETRIC) {       // see spd_ai_pairwise_kernel
        const int64_t per_batch = sym_tiles;
        b = blockIdx.x / per_batch;
        int64_t t = blockIdx.x - b * per_batch;
        cg = 0;
        for (;;) {
            int64_t cnt = ((cg + 1) * (int64_t)blockDim.x + rows - 1) / rows;
            if (cnt > row_chunks) cnt = row_chunks;
            if (t < cnt) break;
            t -= cnt;
            ++cg;
        }
        rc = t;
    } else {
        const int64_t bid = blockIdx.x;
        cg = bid % col_blocks;
        rc = (bid / col_blocks) % row_chunks;
        b = bid / ((int64_t)col_blocks * row_chunks);
    }
    const int64_t j0 = cg * blockDim.x;
    const int64_t j = j0 + threadIdx.x;
    const int64_t jc = j < n2 ? j : n2 - 1;
    const int64_t i0 = rc * rows;
    const int64_t i1 = (i0 + rows < n1) ? i0 + rows : n1;
    const double* Gj = G + b * g_batch_stride + jc;
    const double g00 = Gj[0], g10 = Gj[n2], g11 = Gj[2 * n2];
    const LogTabRegs lr = LogTabRegs::load();
    // the block's rows of W and their log-determinant terms: thread r prepares row i0 + r (rows <= 64), everybody reads them back from LDS
    // with a wave-uniform address (a broadcast: no VALU instruction, no scalar-memory round trip per row)
    __shared__ __attribute__((aligned(16))) double wrow[64 * 4];
    double w0l = 1.0, w1l = 0.0, w2l = 1.0;
    if ((int)threadIdx.x < rows) {
        int64_t ir = i0 + threadIdx.x;
        ir = ir < n1 ? ir : n1 - 1;
        const double* Wr = Winv + b * w_batch_stride + ir * 3;
        w0l = Wr[0], w1l = Wr[1], w2l = Wr[2];
    }
    __syncthreads();                                     // the tables
    if ((int)threadIdx.x < rows) {
        wrow[4 * threadIdx.x + 0] = w0l;
        wrow[4 * threadIdx.x + 1] = w1l;
        wrow[4 * threadIdx.x + 2] = w2l;
        wrow[4 * threadIdx.x + 3] = 2.0 * log_tab(w0l * w2l, lr, ltab);      // 2 log(w00 w11)
    }
    __syncthreads();
    if ((flags & GABO_SYMMETRIC) && j0 + (int64_t)(threadIdx.x | 63) < i0) return;
    double ec[4];
    static_for<4>([&](auto k) { ec[decltype(k)::value] = kExpC256[decltype(k)::value]; });
    double c24 = 1.0 / 24.0, magic = kMagicRound, tiny = 1e-290;
    asm volatile("" : "+v"(c24), "+v"(magic));
    asm volatile("" : "+s"(tiny));
    // 2 (log(w00 w11) + log(g00 g11)) = log det M: the lane's column term and the row's term
    const double lg2 = 2.0 * log_tab(g00 * g11, lr, ltab);
    const double beta_eps = beta * 1e-15;
    double* orow = out + b * n1 * n2 + i0 * n2 + j;
    const int nrows = (int)(i1 - i0);
    // x1 is x2: row r of the block is stored by the lanes with i0 + r <= j
    const int64_t jrel = j - i0;
    const int rmax = j >= n2 ? -1 : ((flags & GABO_SYMMETRIC) ? (jrel < 0 ? -1 : (jrel > 63 ? 63 : (int)jrel)) : 63);
    for (int r = 0; r < nrows; ++r, orow += n2) {
        const double2 wa = *reinterpret_cast<const double2*>(wrow + 4 * r);
        const double2 wb = *reinterpret_cast<const double2*>(wrow + 4 * r + 2);
        const double w0 = wa.x, w1 = wa.y, w2 = wb.x, lw2 = wb.y;
        const double c00 = w0 * g00, c11 = w2 * g11;
        const double c10 = __builtin_fma(w1, g00, w2 * g10);
        const double m00 = c00 * c00, m10d = (c10 + c10) * c00, m11 = __builtin_fma(c10, c10, c11 * c11);
        const double tr = m00 + m11, df = m00 - m11;
        const double x2 = tr + sqrt_nz_cubic(__builtin_fma(df, df, __builtin_fma(m10d, m10d, tiny)));      // 2 lambda_+
        // log(x2 / 2): log_tab with the halving folded into the exponent
        const int ke = __builtin_amdgcn_frexp_exp(x2 * lr.sqrt2);
        const double m = __builtin_ldexp(x2, 1 - ke);
        const unsigned hi = (unsigned)__double2hiint(m);
        const double2 cl = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(ltab) + ((hi >> 9) & 0xff0u));
        const double rr = __builtin_fma(m, cl.x, -1.0);
        double p = MathRegs::fmac(rr, lr.c7, lr.c6);
        p = MathRegs::fmac(p, rr, lr.c5);
        p = MathRegs::fmac(p, rr, lr.c4);
        p = MathRegs::fmac(p, rr, lr.c3);
        p = __builtin_fma(p, rr, -0.5);
        const double lp = __builtin_fma(rr * rr, p, rr);
        const double dk = (double)(ke - 2);
        const double l1 = __builtin_fma(dk, lr.ln2_hi, cl.y) + __builtin_fma(dk, lr.ln2_lo, lp);
        const double l2 = (lg2 - l1) + lw2;
        const double sq = __builtin_fma(l1, l1, l2 * l2);
        // exp(-beta (s + 1e-15)) (spd_utils_torch.py:120, kernels_spd.py:94-98; sqrt(x)^2 = x to an ulp)
        const double y = __builtin_fmin(__builtin_fma(sq, beta, beta_eps), 800.0);
        const double val = exp_of_minus_tab256_magic(y, ec, c24, magic, etab);
        if (r <= rmax) {
            if constexpr (NT) __builtin_nontemporal_store(val, orow);
            else *orow = val;
        }
    }
}

// ---- round 3: d = 2 Gaussian values without a square root or a logarithm per pair --------------------------------------------------
// M = C C^T, C = W G lower triangular:  log lambda_+- = s +- delta  with  s = log sqrt(det M) = log(c00 c11) = log(w00 w11) + log(g00 g11)
// (a sum of per-POINT terms) and delta = acosh(tau),  tau = tr M / (2 sqrt(det M)) = 1 + ((c00 - c11)^2 + c10^2) / (2 c00 c11)  (sqrt(det M) =
// c00 c11 exactly; the numerator has no cancellation; 1 / (c00 c11) is again a product of per-point terms).  Hence
//     d^2 = log^2 lambda_+ + log^2 lambda_- = 2 s^2 + 2 A(tau),   A(tau) = acosh(tau)^2,
// and A is analytic on all of [1, inf) (the square removes acosh's square-root singularity at 1: A = 2u - u^2/3 + ..., u = tau - 1), so it
// comes from a table: slot = low exponent bits + top 4 mantissa bits of tau (one bit-field extract), local variable t = tau minus tau with its lower bits cleared (exact),
// degree 7, coefficients by four ds_read_b128 (csrc/gabo_acosh2_table.hpp, tools/sim/fit_acosh2_table.py: as accurate as the sqrt + log form,
// 2e-15 of max(1, d^2) on the benchmark distribution).  The block's copy of the table is pre-multiplied by -512 beta / ln2 (and carries the
// 1e-15 of spd_utils_torch.py:120 in its constant terms), the per-point logarithms by sqrt(512 beta / ln2): Y = -s'^2 + Y_A is -beta (d^2 +
// 1e-15) in units of ln2/256, whose exp needs no argument reduction products (Y - rint(Y) is exact).  ~36 instructions per pair where the
// sqrt + log form had ~64.  tau >= 2^12 (eigenvalue ratio of M beyond e^18) or NaN: those lanes take acosh = log(tau + sqrt(tau^2 - 1)) from OCML
// behind a wave-uniform branch.
#ifndef GABO_GAUSS2_MAX_ROWS      /* rows of W a block of the d = 2 Gaussian kernel prepares (<= its 256 threads); GABO_PAIR_ROWS_GAUSS2 <= this */
#define GABO_GAUSS2_MAX_ROWS 64
#endif
constexpr double kGauss2L = 0.0027076061740622863;      // ln2 / 256
// [0..3] l^k / k! (k = 1..4), [4] 1.5 2^52, [5] 2^12
__constant__ double kGauss2C[6] = {kGauss2L, kGauss2L * kGauss2L / 2.0, kGauss2L * kGauss2L * kGauss2L / 6.0,
                                   kGauss2L * kGauss2L * kGauss2L * kGauss2L / 24.0, 6755399441055744.0, 4096.0};

template <bool NT, bool SYM>
__global__ __launch_bounds__(256) void spd_ai_gauss2_kernel(const double* __restrict__ Winv, const double* __restrict__ G,
                                                            double* __restrict__ out, int64_t n1, int64_t n2, int64_t w_batch_stride,
                                                            int64_t g_batch_stride, int rows, int col_blocks, int row_chunks,
                                                            int64_t sym_tiles, double beta, int flags) {
#ifndef GABO_GAUSS2_OCML_LOG
    __shared__ __attribute__((aligned(16))) double ltab[512];
#endif
    __shared__ double etab[256];
    __shared__ __attribute__((aligned(16))) double atab[kAcosh2Slots * kAcosh2Stride];
    __shared__ __attribute__((aligned(16))) double wrow[GABO_GAUSS2_MAX_ROWS * 4];
    const int tid = threadIdx.x;
#ifndef GABO_GAUSS2_OCML_LOG
    for (int k = tid; k < 512; k += blockDim.x) ltab[k] = kLogTab[k];
#endif
    for (int k = tid; k < 256; k += blockDim.x) etab[k] = kExp2Tab256[k];
    {
        const double sc = beta * (-512.0 / 0.69314718055994530942), eps = beta * (-256.0 / 0.69314718055994530942) * 1e-15;
        // (rows 176..239 of the table are unused - binades 12..15 - and never read by a lane whose value is kept)
        constexpr int kUsedLo = (kAcosh2Binades - 1) * 16 * kAcosh2Stride, kUsedHi = 240 * kAcosh2Stride;
        for (int k = tid; k < kAcosh2Slots * kAcosh2Stride - (kUsedHi - kUsedLo); k += blockDim.x) {
            const int kk = k < kUsedLo ? k : k + (kUsedHi - kUsedLo);
            const double v = kAcosh2Tab[kk] * sc;
            atab[kk] = (kk % kAcosh2Stride == 0) ? v + eps : v;
        }
    }
    int64_t cg, rc, b;
    if constexpr (SYM) {       // see spd_ai_pairwise_kernel
        const int64_t per_batch = sym_tiles;
        b = blockIdx.x / per_batch;
        int64_t t = blockIdx.x - b * per_batch;
        cg = 0;
        for (;;) {
            int64_t cnt = ((cg + 1) * (int64_t)blockDim.x + rows - 1) / rows;
            if (cnt > row_chunks) cnt = row_chunks;
            if (t < cnt) break;
            t -= cnt;
            ++cg;
        }
        rc = t;
    } else {
        const int64_t bid = blockIdx.x;
        cg = bid % col_blocks;
        rc = (bid / col_blocks) % row_chunks;
        b = bid / ((int64_t)col_blocks * row_chunks);
    }
    const int64_t j0 = cg * blockDim.x;
    const int64_t j = j0 + tid;
    const int64_t jc = j < n2 ? j : n2 - 1;
    const int64_t i0 = rc * rows;
    const int64_t i1 = (i0 + rows < n1) ? i0 + rows : n1;
    const double* Gj = G + b * g_batch_stride + jc;
    const double g00 = Gj[0], g10 = Gj[n2], g11 = Gj[2 * n2];
#ifndef GABO_GAUSS2_OCML_LOG
    const LogTabRegs lr = LogTabRegs::load();
#define GABO_G2_LOG(x) log_tab((x), lr, ltab)
#else
#define GABO_G2_LOG(x) log(x)      /* two logarithms per thread, once: no table to copy into LDS for them */
#endif
    const double sscale = __builtin_sqrt(beta * (512.0 / 0.69314718055994530942));
    // the block's rows of W with their per-point terms: thread r prepares row i0 + r (rows <= 64), everybody reads them back from LDS with a
    // wave-uniform address (a broadcast: no VALU instruction, no scalar-memory round trip per row)
    double w0l = 1.0, w1l = 0.0, w2l = 1.0;
    if (tid < rows) {
        int64_t ir = i0 + tid;
        ir = ir < n1 ? ir : n1 - 1;
        const double* Wr = Winv + b * w_batch_stride + ir * 3;
        w0l = Wr[0], w1l = Wr[1], w2l = Wr[2];
    }
    __syncthreads();                                     // the tables
    // tau = tr M / (2 sqrt(det M)) with M = C C^T, C = W G = [[a, 0], [c, b]]:  tau = 1 + ((a - b)^2 + c^2) / (2 a b)  (no cancellation in the numerator).
    // The quotient is folded into the operands: rows scaled by 1 / sqrt(2 w00 w11), columns by 1 / sqrt(g00 g11) - per-POINT factors - so that
    // with the scaled entries  tau = 1 + (a' - b')^2 + c'^2:  six instructions per pair (round 4: nine, with the product of the two reciprocals
    // formed per pair), tau >= 1 by construction (two FMAs onto 1.0) and NaN operands stay NaN.
    if (tid < rows) {
        const double dw = w0l * w2l;
        const double sw = rsqrt_nz(dw + dw);
        wrow[4 * tid + 0] = w0l * sw;
        wrow[4 * tid + 1] = w1l * sw;
        wrow[4 * tid + 2] = w2l * sw;
        wrow[4 * tid + 3] = sscale * GABO_G2_LOG(dw);      // sqrt(512 beta / ln2) log(w00 w11)
    }
    __syncthreads();
    if (SYM && j0 + (int64_t)(tid | 63) < i0) return;
    double ec[6];
    static_for<6>([&](auto k) { ec[decltype(k)::value] = kGauss2C[decltype(k)::value]; });
    double c4v = ec[3];
    asm volatile("" : "+v"(c4v));
    const double dg = g00 * g11;
    const double bcol = sscale * GABO_G2_LOG(dg);       // the lane's column terms
    const double sg = rsqrt_nz(dg);
    const double q00 = g00 * sg, q10 = g10 * sg, q11 = g11 * sg;       // the column's factor scaled by 1 / sqrt(g00 g11)
    double* orow = out + b * n1 * n2 + i0 * n2 + j;
    const int nrows = __builtin_amdgcn_readfirstlane((int)(i1 - i0));      // (wave-uniform: the loop counter and its compare stay on the scalar unit)
    // x1 is x2: row r of the block is stored by the lanes with i0 + r <= j
    const int64_t jrel = j - i0;
    const int rmax = j >= n2 ? -1 : (SYM ? (jrel < 0 ? -1 : (jrel > GABO_GAUSS2_MAX_ROWS - 1 ? GABO_GAUSS2_MAX_ROWS - 1 : (int)jrel)) : GABO_GAUSS2_MAX_ROWS - 1);
    typedef double g2_v2d __attribute__((ext_vector_type(2)));
    if (!SYM && j >= n2) return;       // (no barrier below; the full build then stores without a per-row predicate)
    for (int r = 0; r < nrows; ++r, orow += n2) {
        const g2_v2d wa = *reinterpret_cast<const g2_v2d*>(wrow + 4 * r);
        const g2_v2d wb = *reinterpret_cast<const g2_v2d*>(wrow + 4 * r + 2);
        const double w0 = wa[0], w1 = wa[1], w2 = wb[0], arow = wb[1];
        const double hh = __builtin_fma(w0, q00, -(w2 * q11));          // a' - b'
        const double cc = __builtin_fma(w1, q00, w2 * q10);             // c'
        const double tau = __builtin_fma(cc, cc, __builtin_fma(hh, hh, 1.0));
        const unsigned hi = (unsigned)__double2hiint(tau);
        // row of the table: bits 16..23 of the high word (in range by construction: one v_bfe_u32, no bias subtraction, and the four reads
        // share one address register with immediate offsets)
        static_assert(kAcosh2MBits == 4 && kAcosh2Slots == 256, "row index = 8 bits");
        const unsigned off = __umul24((hi >> 16) & 0xffu, (unsigned)(kAcosh2Stride * sizeof(double)));
        const double t = tau - __hiloint2double((int)(hi & ~((1u << (20 - kAcosh2MBits)) - 1u)), 0);
        const g2_v2d* row = reinterpret_cast<const g2_v2d*>(reinterpret_cast<const char*>(atab) + off);
        static_assert(kAcosh2Deg == 7, "four coefficient pairs");
        const g2_v2d c67 = row[3], c45 = row[2], c23 = row[1], c01 = row[0];
        double ya = __builtin_fma(c67[1], t, c67[0]);
        ya = __builtin_fma(ya, t, c45[1]);
        ya = __builtin_fma(ya, t, c45[0]);
        ya = __builtin_fma(ya, t, c23[1]);
        ya = __builtin_fma(ya, t, c23[0]);
        ya = __builtin_fma(ya, t, c01[1]);
        ya = __builtin_fma(ya, t, c01[0]);
        if (__builtin_expect(__any(!(tau < ec[5])), 0)) {
            if (!(tau < ec[5])) {       // beyond the table (or NaN): acosh from OCML
                const double dl = log(tau + __builtin_sqrt(__builtin_fma(tau, tau, -1.0)));
                ya = -(sscale * sscale) * __builtin_fma(dl, dl, 0.5e-15);
            }
        }
        const double sp = arow + bcol;
        const double y = __builtin_fma(-sp, sp, ya);           // -beta (d^2 + 1e-15) 256 / ln2  (spd_utils_torch.py:120, kernels_spd.py:94-98)
        const double km = y + ec[4];
        const double kk = km - ec[4];
        const double rr = y - kk;
        double p = __builtin_fma(rr, c4v, ec[2]);
        p = __builtin_fma(p, rr, ec[1]);
        p = __builtin_fma(p, rr, ec[0]);
        p = p * rr;
        const int ki = __double2loint(km);
        const double e = etab[ki & 255];
        const double val = __builtin_ldexp(__builtin_fma(e, p, e), ki >> 8);
        if (!SYM || r <= rmax) {
            if constexpr (NT) __builtin_nontemporal_store(val, orow);
            else *orow = val;
        }
    }
}
#undef GABO_G2_LOG

// prepared: the workspace already holds the factors (W for b1 * n1 matrices, then G: the fused projection of the nested kernels wrote them,
// nested_spd_gram.hip); s1 / s2 then only say whether a set is shared across the batch (0) or not
template <int D>
static int launch_spd_ai(const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                         int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st, bool prepared = false) {
    constexpr int T = tri_size(D);
    const int64_t b1 = (s1 == 0) ? 1 : batch;  // a shared set is factored once
    const int64_t b2 = (s2 == 0) ? 1 : batch;
    double* W = ws;
    double* G = ws + b1 * n1 * T;
    if (!prepared) launch_spd_prep<D>(x1, x2, W, G, b1, b2, n1, n2, s1, s2, status, st, /*lenient2=*/1);
    // tile shape: 64..256 columns per block, `rows` rows per block; keep >= ~8 blocks per CU when the problem allows
    // (d = 10, N = 4096, 8 rows: 256 / 128 / 64 threads per block = 2.57 / 2.56 / 2.55 ms - one-wave blocks: a block's slots are not held until its slowest wave ends; d = 7: 1.25 -> 1.23 ms, d <= 5: no gain)
#ifndef GABO_PAIR_THREADS
#define GABO_PAIR_THREADS (D >= 7 ? 64 : 256)
#endif
    int threads = n2 >= 256 ? GABO_PAIR_THREADS : (n2 > 128 ? 192 : (n2 > 64 ? 128 : 64));
    int64_t col_blocks = (n2 + threads - 1) / threads;
    // rows per block: a wave's prologue (log table into LDS, for d = 2 the per-point log terms) is amortised over them; d = 2 pairs are ~75
    // instructions; measured at N = 4096, d = 2: 64 / 32 / 16 / 8 / 4 rows per block = 66 / 61 / 59 / 63 / 71 us
#ifndef GABO_PAIR_ROWS_D2
#define GABO_PAIR_ROWS_D2 16
#endif
    // d = 10, N = 4096 (blocks vary in duration with their pairs' QL iteration counts: smaller blocks balance the tail, larger ones amortise
    // the prologue): 64 / 32 / 16 / 8 / 4 / 2 rows = 2.72 / 2.64 / 2.60 / 2.58 / 2.58 / 2.62 ms
#ifndef GABO_PAIR_ROWS
#define GABO_PAIR_ROWS (D >= 9 ? 8 : 16)
#endif
    // the d = 2 Gaussian kernel copies a 20 KB table into LDS per block: 64 rows per block amortise it (N = 4096: 16 / 32 / 64 rows = 46.5 / 42.4 /
    // 40.7 us including the prep launch)
#ifndef GABO_PAIR_ROWS_GAUSS2
#define GABO_PAIR_ROWS_GAUSS2 64
#endif
    const bool gauss2 = D == 2 && (flags & GABO_OUT_MASK) == GABO_OUT_GAUSSIAN && !dist_out && beta >= 0.0;
    int rows = D == 2 ? (gauss2 ? GABO_PAIR_ROWS_GAUSS2 : GABO_PAIR_ROWS_D2) : GABO_PAIR_ROWS;
    while (rows > 1 && col_blocks * ((n1 + rows - 1) / rows) * batch < (rows > 16 ? 1024 : 4096)) rows >>= 1;
    int64_t row_chunks = (n1 + rows - 1) / rows;
    int64_t sym_tiles = 0;
    if (flags & GABO_SYMMETRIC) {
        for (int64_t cg = 0; cg < col_blocks; ++cg) {
            int64_t cnt = ((cg + 1) * threads + rows - 1) / rows;
            sym_tiles += cnt < row_chunks ? cnt : row_chunks;
        }
    }
    int64_t nblocks = ((flags & GABO_SYMMETRIC) ? sym_tiles : col_blocks * row_chunks) * batch;
    if (nblocks > 0x7fffffffLL) return GABO_ERR_ARG;
    bool special2 = false;
#ifndef GABO_PAIR_NO_GAUSS2
    if constexpr (D == 2) special2 = gauss2 && rows <= GABO_GAUSS2_MAX_ROWS;
#endif
    if (special2) {
#ifdef GABO_GAUSS2_NO_NT      /* A/B: plain stores whatever the size */
        const bool streaming = false;
#else
        const bool streaming = batch * n1 * n2 * 8 > (32ll << 20) && !(flags & GABO_SYMMETRIC);      // beyond the L2 caches (see sphere_pairwise.hip)
#endif
#ifdef GABO_GAUSS2_LOG
#define GABO_GAUSS2_KERNEL spd_ai_gauss2_log_kernel
#else
#define GABO_GAUSS2_KERNEL spd_ai_gauss2_kernel
#endif
#define GABO_GAUSS2_LAUNCH(...)                                                                                                     \
        hipLaunchKernelGGL((GABO_GAUSS2_KERNEL<__VA_ARGS__>), dim3((unsigned)nblocks), dim3(threads), 0, st, W, G, out, n1, n2,             \
                           (s1 == 0) ? (int64_t)0 : n1 * T, (s2 == 0) ? (int64_t)0 : n2 * T, rows, (int)col_blocks, (int)row_chunks,       \
                           sym_tiles, beta, flags)
#ifdef GABO_GAUSS2_LOG
        if (streaming) GABO_GAUSS2_LAUNCH(true);
        else GABO_GAUSS2_LAUNCH(false);
#else
        if (flags & GABO_SYMMETRIC) GABO_GAUSS2_LAUNCH(false, true);        // (never streamed: see `streaming`)
        else if (streaming) GABO_GAUSS2_LAUNCH(true, false);
        else GABO_GAUSS2_LAUNCH(false, false);
#endif
#undef GABO_GAUSS2_LAUNCH
    } else
    hipLaunchKernelGGL((spd_ai_pairwise_kernel<D>), dim3((unsigned)nblocks), dim3(threads), 0, st, W, G, out, dist_out, n1, n2,
                       (s1 == 0) ? (int64_t)0 : n1 * T, (s2 == 0) ? (int64_t)0 : n2 * T, rows, (int)col_blocks,
                       (int)row_chunks, sym_tiles, beta, flags);
    if (flags & GABO_SYMMETRIC) {
        int tiles = (int)((n1 + 31) / 32);
        hipLaunchKernelGGL((mirror_upper_kernel<0>), dim3((unsigned)((int64_t)tiles * (tiles + 1) / 2), (unsigned)batch), dim3(256), 0, st,
                           out, n1, tiles);
        if (dist_out)
            hipLaunchKernelGGL((mirror_upper_kernel<0>), dim3((unsigned)((int64_t)tiles * (tiles + 1) / 2), (unsigned)batch), dim3(256), 0,
                               st, dist_out, n1, tiles);
    }
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // namespace gabo

namespace gabo {
// the pairwise launch alone on factors already in the workspace, d = 2 ... 4 (spd_pairwise.hip; called by nested_spd_gram.hip)
int launch_spd_ai_prepared(int d, double* out, int64_t batch, int64_t n1, int64_t n2, bool shared1, bool shared2, double beta, int flags,
                           double* ws, hipStream_t st);
// dimensions 13..16 are instantiated in their own translation unit (spd_pairwise_wide.hip), 17..20 in two more
int launch_spd_ai_wide2(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                        int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st);
int launch_spd_ai_wide3(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                        int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st);
int launch_spd_ai_wide(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                       int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st);
}  // namespace gabo
