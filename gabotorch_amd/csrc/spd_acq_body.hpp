// Device-side body of the fused acquisition evaluation (see spd_acq.hip for the description): shared by the stand-alone kernel
// and by the trust-region iteration kernel (spd_tr.hip).
#pragma once
#include "gabo_device.hpp"
#include "spd_prep.hpp"
#include "spd_jacobi.hpp"
#include "spd_eigvec.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

static __device__ __forceinline__ double wave_sum64(double v) { return wave_allsum(v); }      // gabo_device.hpp

using AcqParams = gabo_spd_acq_params;   // include/gabo_hip.h

// sum_{j < n} col[j * n] * x[j]: one output of the two triangular matrix-vector products of the exact-GP posterior per lane.  L^-1 is
// stored dense with exact zeros above the diagonal, so the sum runs over ALL j (uniform trip count, nothing to mask); eight loads are
// issued before the FMAs that use them and four partial sums keep the FMA chain short.  (The plain triangular loop with its lane-dependent
// bound waited for one load per term: 300 cycles per term from L2, 150 once the factors were staged in LDS; a version that masked the
// terms outside the triangle instead of using the zeros was slower still - tools/tr_clocks.py.)
static __device__ __forceinline__ double strided_dot(const double* __restrict__ col, const double* __restrict__ x, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        double av[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = col[(j + u) * n];
            xv[u] = x[j + u];
        }
        s0 = __builtin_fma(av[0], xv[0], s0); s1 = __builtin_fma(av[1], xv[1], s1);
        s2 = __builtin_fma(av[2], xv[2], s2); s3 = __builtin_fma(av[3], xv[3], s3);
        s0 = __builtin_fma(av[4], xv[4], s0); s1 = __builtin_fma(av[5], xv[5], s1);
        s2 = __builtin_fma(av[6], xv[6], s2); s3 = __builtin_fma(av[7], xv[7], s3);
    }
    for (; j < n; ++j) s0 = __builtin_fma(col[j * n], x[j], s0);
    return (s0 + s1) + (s2 + s3);
}

// LDS needed by acq_eval<D>: static part (doubles) + 3 n doubles of dynamic scratch
template <int D>
struct AcqLds {
    static constexpr int T = tri_size(D);
    double acc[T * 64];
    double vls[(D <= 8) ? 2 : D * D * 64];       // eigenvector columns of the lanes: only when they do not live in registers (D > 8)
    double red[T];
    double wl[T];
};

// One wave: acquisition value and (grad != nullptr) Euclidean gradient, Mandel, at the SPD point xrow (Mandel, global or LDS).
// F: T * n doubles of global scratch owned by this wave; dyn: 3 n doubles of LDS.  All 64 lanes call it; ends with the outputs
// written by the owning lanes (no trailing barrier).
#ifdef GABO_ACQ_NOINLINE        /* A/B (round 6): one out-of-line copy per translation unit instead of one inlined copy per call site */
#define GABO_ACQ_INLINE __attribute__((noinline))
#else
#define GABO_ACQ_INLINE __forceinline__
#endif
template <int D>
__device__ GABO_ACQ_INLINE void acq_eval(const double* __restrict__ xrow, const AcqParams& P, double* __restrict__ value_out,
                                         double* __restrict__ grad_out, double* __restrict__ F, AcqLds<D>& L, double* dyn,
                                         int* __restrict__ status, int64_t index) {
    constexpr int T = tri_size(D);
    constexpr int LD = 64;
    const double* __restrict__ G = P.train_factors;
    const double* __restrict__ alpha = P.alpha;
    const double* __restrict__ linv = P.linv;
    const double* __restrict__ linv_t = P.linv_t;
    const int64_t n = P.n;
    const double beta = P.beta, mean0 = P.mean, os = P.outputscale, kxx = P.kxx, best_f = P.best_f, out_sign = P.out_sign;
    const int mode = P.flags & GABO_OUT_MASK, kind = P.kind, maximize = P.maximize;
    double* acc = L.acc;
    double* vls = L.vls;
    double* red = L.red;
    double* wl = L.wl;
    double* ks = dyn;          // n : outputscale * k_j
    double* kd = ks + n;       // n : d k_j / d (d_j^2)
    double* vv = kd + n;       // n : L^-1 ks
    const int lane = threadIdx.x;
    const int64_t i = index;
    const bool want_grad = grad_out != nullptr;
    GABO_TICK(100);
    // ---- candidate side: W = chol(x*)^-1, computed by every lane (wave-uniform, a few hundred flops)
    // and kept in LDS (broadcast reads with compile-time offsets) so that it does not compete with M for registers
    {
        double a[T], w[T];
        const bool bad = mandel_cholesky<D>(xrow, a);
        if (bad && lane == 0 && status) {
            if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)i;
        }
        lower_inverse<D>(a, w);
        if (lane == 0) static_for<T>([&](auto ee) { wl[decltype(ee)::value] = w[decltype(ee)::value]; });
    }
    __syncthreads();
    GABO_TICK(101);
    const double* w = wl;
    const LogRegs logc = LogRegs::load();
    // An experiment of round 6, off by default (-DGABO_ACQ_COMPACT): for D <= 8 and n <= 64 logm(M_j) stays in the lane's registers until its weight
    // is known and the weighted sum over the lanes is a wave reduction in registers, instead of the spill to F, the LDS column per lane and the
    // reduction by T lanes reading 64 entries each.  The instrumented build showed the two phases 4.2 k + 3.4 k -> 1.0 k + 4.0 k cycles, the product
    // build was 23 % SLOWER over the whole solve.
#ifdef GABO_ACQ_COMPACT          /* A/B (round 6): measured SLOWER - the single-launch solve at d = 5, n = 50 went from 781 to 964 us (rocprofv3) with it: */
    constexpr bool kCompact = D <= 8;      /* fifteen more doubles live across the posterior phase of a kernel that already fills 512 registers */
#else
    constexpr bool kCompact = false;
#endif
    const bool compact = kCompact && want_grad && n <= 64;
    double freg[kCompact ? T : 1];
    // L^-1 and L^-T the same pointer: the caller handed over the symmetric A = (outputscale K + noise I)^-1 instead (gabo_gp_factor's `kinv`):
    // var = k** - ks^T A ks and the gradient's L^-T L^-1 ks are then ONE matrix-vector product u = A ks instead of two triangular ones
    const bool sym_inverse = linv != nullptr && linv == linv_t;
    for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool live = j < n;
        const double* Gj = G + (live ? j : n - 1);
        double m[T];
        static_for<T>([&](auto ee) { m[decltype(ee)::value] = 0.0; });
        static_for<D>([&](auto cc) {
            constexpr int col = decltype(cc)::value;
            double g[D - col], c[D - col];
            static_for<D - col>([&](auto kk) { g[decltype(kk)::value] = Gj[(int64_t)tri(col + decltype(kk)::value, col) * n]; });
            static_for<D - col>([&](auto rr) {
                constexpr int r = col + decltype(rr)::value;
                double a = w[tri(r, col)] * g[0];
                static_for<r - col>([&](auto kk) {
                    constexpr int k = col + 1 + decltype(kk)::value;
                    a = __builtin_fma(w[tri(r, k)], g[k - col], a);
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
        // eigenvectors: registers for D <= 8 (this kernel runs 1-2 waves per SIMD: LDS round trips would be exposed latency),
        // the lane's LDS column above that
        // (registers: Householder + QL with vectors, spd_eigvec.hpp - a far shorter dependent chain than the cyclic Jacobi it replaced,
        // which was 47 % of an evaluation at one wave per SIMD)
        constexpr bool kRegV = D <= 8;
        double* vl = vls + lane;
        double vreg[kRegV ? D * D : 1];
        double lam[D];
        if constexpr (kRegV) {
            // (value only: the same eigenvalues, bit for bit, without the eigenvectors - a third of the eigen-solver's instructions)
#ifdef GABO_ACQ_TWO_PASS       /* A/B (round 6): the backward's two-pass decomposition here too - measured, see DESIGN.md */
            if (want_grad) sym_eig_reg_two_pass<D>(m, lam, vreg);
#else
            if (want_grad) sym_eig_reg<D>(m, lam, vreg);
#endif
            else sym_eig_reg_values<D>(m, lam);
        } else {
            jacobi_eig<D>(m, vl);
            static_for<D>([&](auto kk) { lam[decltype(kk)::value] = m[tri(decltype(kk)::value, decltype(kk)::value)]; });
        }
        auto Vat = [&](int r, int c) -> double { if constexpr (kRegV) return vreg[r * D + c]; else return vl[(r * D + c) * 64]; };
        double lg[D];
        double s = 0.0;
        static_for<D>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            lg[k] = log_pos(lam[k], logc);                // (fdlibm scheme, 1 ulp, ~35 instructions; OCML's log is ~100: D of them per pair)
            s = __builtin_fma(lg[k], lg[k], s);
        });
        const double d2 = s + 1e-15;                      // spd_utils_torch.py:120
        const double dist = __builtin_sqrt(d2);
        double kj, dk;
        if (mode == GABO_OUT_GAUSSIAN) {
            kj = exp(-((dist * dist) * beta));
            dk = -beta * kj;
        } else {
            kj = exp(-(dist * beta));
            dk = -beta * kj / (2.0 * dist);
        }
        if (live) {
            ks[j] = os * kj;
            kd[j] = dk;
            if (want_grad) {
                static_for<D>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    static_for<r + 1>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        double f = 0.0;
                        static_for<D>([&](auto kk) {
                            constexpr int k = decltype(kk)::value;
                            f = __builtin_fma(Vat(r, k) * lg[k], Vat(c, k), f);
                        });
                        if constexpr (kCompact) {
                            if (compact) freg[tri(r, c)] = f;
                            else F[(int64_t)tri(r, c) * n + j] = f;
                        } else {
                            F[(int64_t)tri(r, c) * n + j] = f;
                        }
                    });
                });
            }
        }
    }
    __syncthreads();
    GABO_TICK(102);
    // ---- exact-GP posterior + acquisition (same arithmetic as gp_acquisition.hip)
    double part = 0.0;
    for (int64_t j = lane; j < n; j += 64) part = __builtin_fma(ks[j], alpha[j], part);
    const double mean = mean0 + wave_sum64(part);
    const double sgn = maximize ? 1.0 : -1.0;
    double g_mean, g_var = 0.0;
    if (kind == GABO_ACQ_POSTERIOR_MEAN) {
        if (lane == 0) *value_out = out_sign * sgn * mean;
        g_mean = sgn;
    } else {
        part = 0.0;
        if (sym_inverse) {
            for (int64_t r = lane; r < n; r += 64) {
                vv[r] = strided_dot(linv + r, ks, (int)n);        // u_r = (A ks)_r  (A symmetric: its column r)
                part = __builtin_fma(ks[r], vv[r], part);
            }
        } else {
            for (int64_t r = lane; r < n; r += 64) {
                vv[r] = strided_dot(linv_t + r, ks, (int)n);      // (L^-1 ks)_r = sum_j L^-T[j][r] ks[j]  (zero for j > r)
                part = __builtin_fma(vv[r], vv[r], part);
            }
        }
        const double var = os * kxx - wave_sum64(part);
        const bool clamped = !(var > 1e-9);
        const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
        const double u = sgn * (mean - best_f) / sigma;
        const double pdf = exp(-0.5 * u * u) * 0.3989422804014327;
        const double cdf = 0.5 * (1.0 + erf(u * 0.7071067811865476));
        if (lane == 0) *value_out = out_sign * sigma * (pdf + u * cdf);
        g_mean = sgn * cdf;
        g_var = clamped ? 0.0 : 0.5 * pdf / sigma;
    }
    if (!want_grad) return;
    __syncthreads();
    GABO_TICK(103);
    // ---- weights w_j = d(out_sign acq)/d(d_j^2) and S = sum_j w_j logm(M_j)
    bool reduced = false;
    if constexpr (kCompact) {
        if (compact) {
            double wj = 0.0;
            if (lane < n) {
                double ws = 0.0;
                if (kind != GABO_ACQ_POSTERIOR_MEAN) ws = sym_inverse ? vv[lane] : strided_dot(linv + lane, vv, (int)n);
                wj = (out_sign * os * (g_mean * alpha[lane] - 2.0 * g_var * ws)) * kd[lane];
            }
            GABO_TICK(104);
            static_for<T>([&](auto ee) {
                constexpr int e = decltype(ee)::value;
                const double sume = wave_sum64(lane < n ? wj * freg[e] : 0.0);
                if (lane == 0) red[e] = sume;
            });
            reduced = true;
        }
    }
    if (!reduced) {
    // (one accumulator column per lane in LDS)
    static_for<T>([&](auto ee) { acc[decltype(ee)::value * LD + lane] = 0.0; });
    for (int64_t j = lane; j < n; j += 64) {
        double ws = 0.0;
        if (kind != GABO_ACQ_POSTERIOR_MEAN) ws = sym_inverse ? vv[j] : strided_dot(linv + j, vv, (int)n);     // (L^-T v)_j = sum_r L^-1[r][j] v[r]  (zero for r < j)
        const double gk = out_sign * os * (g_mean * alpha[j] - 2.0 * g_var * ws);
        const double wj = gk * kd[j];
        static_for<T>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            acc[e * LD + lane] = __builtin_fma(wj, F[(int64_t)e * n + j], acc[e * LD + lane]);
        });
    }
    __syncthreads();
    GABO_TICK(104);
    for (int e = lane; e < T; e += 64) {
        // four partial sums, the loop unrolled: the 64 LDS reads are independent and go out back to back (as one running sum the loop
        // waited for each read: 75 cycles per term, 4.8 k of an evaluation's 42 k - tools/tr_clocks.py)
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
        for (int l = 0; l < 64; l += 4) {
            t0 += acc[e * LD + ((l + e) & 63)];
            t1 += acc[e * LD + ((l + 1 + e) & 63)];
            t2 += acc[e * LD + ((l + 2 + e) & 63)];
            t3 += acc[e * LD + ((l + 3 + e) & 63)];
        }
        red[e] = (t0 + t1) + (t2 + t3);
    }
    }
    __syncthreads();
    GABO_TICK(105);
    // grad = -2 W^T S W, Mandel  (as in spd_backward.hip)
#ifdef GABO_ACQ_TAIL_LOOPS      /* A/B: rounds 1-6a - every output entry by one lane as a double loop with lane-dependent bounds: 25 dependent LDS round trips,
                                   7.3 k of an evaluation's 40 k cycles at d = 5 (tools/tr_clocks.py: what remains of "two acquisition evaluations" after its phases) */
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
        grad_out[mandel_pos(D, a, bb)] = (a == bb) ? t : t * kSqrt2;
    }
#else
    // The same sums in the same order as two phases with one entry per lane, the loops unrolled with their lane-dependent bounds as selects, so that the
    // LDS reads of a phase go out together: U[r][b] = sum_{c >= b} S[r][c] W[c][b] for all (r, b), then -2 sum_{r >= a} W[r][a] U[r][b].
    // U goes where the accumulator columns were (the reduction above has consumed them).
    double* U = acc;
    for (int idx = lane; idx < D * D; idx += 64) {
        const int r = idx / D, b = idx - r * D;
        double inner = 0.0;
        static_for<D>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const double srs = r >= c ? red[tri(r, c)] : red[tri(c, r)];
            const double wcb = wl[c >= b ? tri(c, b) : 0];
            inner = c >= b ? __builtin_fma(srs, wcb, inner) : inner;
        });
        U[idx] = inner;
    }
    __syncthreads();
    for (int e = lane; e < T; e += 64) {
        int a = 0;
        while (tri(a + 1, 0) <= e) ++a;
        const int bb = e - tri(a, 0);
        double t = 0.0;
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const double wra = wl[r >= a ? tri(r, a) : 0];
            const double u = U[r * D + bb];
            t = r >= a ? __builtin_fma(wra, u, t) : t;
        });
        t *= -2.0;
        grad_out[mandel_pos(D, a, bb)] = (a == bb) ? t : t * kSqrt2;
    }
#endif
}

// ---- acq_eval<D> split between TWO waves (the two-wave trust-region solve, spd_tr_duo_body.hpp: used while one of its waves would otherwise idle) ---------------
// The evaluation is one dependent chain per lane: M_j -> eigen-decomposition with vectors (13 k cycles at D = 5) -> logm(M_j) -> posterior -> weights -> weighted sum ->
// congruence.  But the posterior and the weights need the EIGENVALUES only (a third of the eigen-solver), and logm(M_j) needs nothing of the posterior:
//   helper wave:  acq_eval_values    M_j, eigenvalues (sym_eig_reg_values: the bits of the full solver), kernel values, posterior, acquisition VALUE, weights w_j -> wts[n]
//   owner wave:   acq_eval_vectors   M_j, eigen-decomposition, logm(M_j) -> F           ... barrier (the caller's) ...
//                 acq_eval_finish    S = sum_j w_j logm(M_j), gradient = -2 W^T S W
// The statements are acq_eval's, in its order, so value and gradient are its bits; each wave has its own AcqLds / 3 n doubles / W.  D <= 8, n <= 64 lanes per pass as there.
template <int D>
__device__ __forceinline__ void acq_eval_pair_matrix(const double* __restrict__ Gj, int64_t n, const double* __restrict__ w, double (&m)[tri_size(D)]) {
    constexpr int T = tri_size(D);
    static_for<T>([&](auto ee) { m[decltype(ee)::value] = 0.0; });
    static_for<D>([&](auto cc) {
        constexpr int col = decltype(cc)::value;
        double g[D - col], c[D - col];
        static_for<D - col>([&](auto kk) { g[decltype(kk)::value] = Gj[(int64_t)tri(col + decltype(kk)::value, col) * n]; });
        static_for<D - col>([&](auto rr) {
            constexpr int r = col + decltype(rr)::value;
            double a = w[tri(r, col)] * g[0];
            static_for<r - col>([&](auto kk) {
                constexpr int k = col + 1 + decltype(kk)::value;
                a = __builtin_fma(w[tri(r, k)], g[k - col], a);
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
}

template <int D>
__device__ __forceinline__ void acq_eval_candidate_side(const double* __restrict__ xrow, AcqLds<D>& L, int* __restrict__ status, int64_t index) {
    constexpr int T = tri_size(D);
    double a[T], w[T];
    const bool bad = mandel_cholesky<D>(xrow, a);
    if (bad && threadIdx.x == 0 && status) {
        if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)index;
    }
    lower_inverse<D>(a, w);
    if (threadIdx.x == 0) static_for<T>([&](auto ee) { L.wl[decltype(ee)::value] = w[decltype(ee)::value]; });
    __syncthreads();
}

template <int D>
__device__ __forceinline__ void acq_eval_values(const double* __restrict__ xrow, const AcqParams& P, double* __restrict__ value_out, double* __restrict__ wts,
                                                AcqLds<D>& L, double* dyn, int* __restrict__ status, int64_t index) {
    static_assert(D <= 8, "register eigen-solver");
    constexpr int T = tri_size(D);
    const double* __restrict__ G = P.train_factors;
    const double* __restrict__ alpha = P.alpha;
    const double* __restrict__ linv = P.linv;
    const double* __restrict__ linv_t = P.linv_t;
    const int64_t n = P.n;
    const double beta = P.beta, mean0 = P.mean, os = P.outputscale, kxx = P.kxx, best_f = P.best_f, out_sign = P.out_sign;
    const int mode = P.flags & GABO_OUT_MASK, kind = P.kind, maximize = P.maximize;
    double* ks = dyn;
    double* kd = ks + n;
    double* vv = kd + n;
    const int lane = threadIdx.x;
    acq_eval_candidate_side<D>(xrow, L, status, index);
    const double* w = L.wl;
    const LogRegs logc = LogRegs::load();
    const bool sym_inverse = linv != nullptr && linv == linv_t;
    for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool live = j < n;
        double m[T];
        acq_eval_pair_matrix<D>(G + (live ? j : n - 1), n, w, m);
        double lam[D];
        sym_eig_reg_values<D>(m, lam);
        double s = 0.0;
        static_for<D>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const double lgk = log_pos(lam[k], logc);
            s = __builtin_fma(lgk, lgk, s);
        });
        const double d2 = s + 1e-15;
        const double dist = __builtin_sqrt(d2);
        double kj, dk;
        if (mode == GABO_OUT_GAUSSIAN) {
            kj = exp(-((dist * dist) * beta));
            dk = -beta * kj;
        } else {
            kj = exp(-(dist * beta));
            dk = -beta * kj / (2.0 * dist);
        }
        if (live) {
            ks[j] = os * kj;
            kd[j] = dk;
        }
    }
    __syncthreads();
    double part = 0.0;
    for (int64_t j = lane; j < n; j += 64) part = __builtin_fma(ks[j], alpha[j], part);
    const double mean = mean0 + wave_sum64(part);
    const double sgn = maximize ? 1.0 : -1.0;
    double g_mean, g_var = 0.0;
    if (kind == GABO_ACQ_POSTERIOR_MEAN) {
        if (lane == 0) *value_out = out_sign * sgn * mean;
        g_mean = sgn;
    } else {
        part = 0.0;
        if (sym_inverse) {
            for (int64_t r = lane; r < n; r += 64) {
                vv[r] = strided_dot(linv + r, ks, (int)n);
                part = __builtin_fma(ks[r], vv[r], part);
            }
        } else {
            for (int64_t r = lane; r < n; r += 64) {
                vv[r] = strided_dot(linv_t + r, ks, (int)n);
                part = __builtin_fma(vv[r], vv[r], part);
            }
        }
        const double var = os * kxx - wave_sum64(part);
        const bool clamped = !(var > 1e-9);
        const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
        const double u = sgn * (mean - best_f) / sigma;
        const double pdf = exp(-0.5 * u * u) * 0.3989422804014327;
        const double cdf = 0.5 * (1.0 + erf(u * 0.7071067811865476));
        if (lane == 0) *value_out = out_sign * sigma * (pdf + u * cdf);
        g_mean = sgn * cdf;
        g_var = clamped ? 0.0 : 0.5 * pdf / sigma;
    }
    __syncthreads();
    for (int64_t j = lane; j < n; j += 64) {
        double ws = 0.0;
        if (kind != GABO_ACQ_POSTERIOR_MEAN) ws = sym_inverse ? vv[j] : strided_dot(linv + j, vv, (int)n);
        const double gk = out_sign * os * (g_mean * alpha[j] - 2.0 * g_var * ws);
        wts[j] = gk * kd[j];
    }
    __syncthreads();
}

template <int D>
__device__ __forceinline__ void acq_eval_vectors(const double* __restrict__ xrow, const AcqParams& P, double* __restrict__ F, AcqLds<D>& L,
                                                 int* __restrict__ status, int64_t index) {
    static_assert(D <= 8, "register eigen-solver");
    constexpr int T = tri_size(D);
    const double* __restrict__ G = P.train_factors;
    const int64_t n = P.n;
    const int lane = threadIdx.x;
    acq_eval_candidate_side<D>(xrow, L, nullptr, index);       // (the helper wave reports a candidate that is not positive definite)
    const double* w = L.wl;
    const LogRegs logc = LogRegs::load();
    for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool live = j < n;
        double m[T];
        acq_eval_pair_matrix<D>(G + (live ? j : n - 1), n, w, m);
        double vreg[D * D], lam[D], lg[D];
        sym_eig_reg<D>(m, lam, vreg);
        static_for<D>([&](auto kk) { lg[decltype(kk)::value] = log_pos(lam[decltype(kk)::value], logc); });
        if (live) {
            static_for<D>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<r + 1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    double f = 0.0;
                    static_for<D>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
                        f = __builtin_fma(vreg[r * D + k] * lg[k], vreg[c * D + k], f);
                    });
                    F[(int64_t)tri(r, c) * n + j] = f;
                });
            });
        }
    }
    __syncthreads();
}

template <int D>
__device__ __forceinline__ void acq_eval_finish(const AcqParams& P, double* __restrict__ grad_out, const double* __restrict__ F, const double* __restrict__ wts,
                                                AcqLds<D>& L) {
    constexpr int T = tri_size(D);
    constexpr int LD = 64;
    const int64_t n = P.n;
    const int lane = threadIdx.x;
    double* acc = L.acc;
    double* red = L.red;
    const double* wl = L.wl;
    static_for<T>([&](auto ee) { acc[decltype(ee)::value * LD + lane] = 0.0; });
    for (int64_t j = lane; j < n; j += 64) {
        const double wj = wts[j];
        static_for<T>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            acc[e * LD + lane] = __builtin_fma(wj, F[(int64_t)e * n + j], acc[e * LD + lane]);
        });
    }
    __syncthreads();
    for (int e = lane; e < T; e += 64) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
        for (int l = 0; l < 64; l += 4) {
            t0 += acc[e * LD + ((l + e) & 63)];
            t1 += acc[e * LD + ((l + 1 + e) & 63)];
            t2 += acc[e * LD + ((l + 2 + e) & 63)];
            t3 += acc[e * LD + ((l + 3 + e) & 63)];
        }
        red[e] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
    double* U = acc;
    for (int idx = lane; idx < D * D; idx += 64) {
        const int r = idx / D, b = idx - r * D;
        double inner = 0.0;
        static_for<D>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const double srs = r >= c ? red[tri(r, c)] : red[tri(c, r)];
            const double wcb = wl[c >= b ? tri(c, b) : 0];
            inner = c >= b ? __builtin_fma(srs, wcb, inner) : inner;
        });
        U[idx] = inner;
    }
    __syncthreads();
    for (int e = lane; e < T; e += 64) {
        int a = 0;
        while (tri(a + 1, 0) <= e) ++a;
        const int bb = e - tri(a, 0);
        double t = 0.0;
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const double wra = wl[r >= a ? tri(r, a) : 0];
            const double u = U[r * D + bb];
            t = r >= a ? __builtin_fma(wra, u, t) : t;
        });
        t *= -2.0;
        grad_out[mandel_pos(D, a, bb)] = (a == bb) ? t : t * kSqrt2;
    }
}

// Same contract as acq_eval<D> for the two Frobenius-type surrogates (kernels_spd.py:190-313):
//   METRIC 1: log-Euclidean  k_j = exp(-beta ||logm x - logm X_j + 1e-15||_F^2),  P.train_factors = Mandel(logm X_j), entry-major [T][n]
//   METRIC 2: Frobenius      k_j = exp(-beta ||x - X_j + 1e-15||_F^2),            P.train_factors = Mandel(X_j),      entry-major [T][n]
// The candidate's eigen-decomposition (registers, every lane redundantly: D <= 8) serves both logm(x) and the adjoint of its
// Frechet derivative (Daleckii-Krein) that carries the gradient from log space back to x.  dyn: 3 n doubles of LDS.
template <int D, int METRIC>
__device__ __forceinline__ void acq_eval_frob(const double* __restrict__ xrow, const AcqParams& P, double* __restrict__ value_out,
                                              double* __restrict__ grad_out, double* dyn) {
    static_assert(D <= 8, "register-resident eigenvectors");
    constexpr int T = tri_size(D);
    const double* __restrict__ Ft = P.train_factors;
    const double* __restrict__ alpha = P.alpha;
    const double* __restrict__ linv = P.linv;
    const double* __restrict__ linv_t = P.linv_t;
    const bool sym_inverse = linv != nullptr && linv == linv_t;
    const int64_t n = P.n;
    const double beta = P.beta, mean0 = P.mean, os = P.outputscale, kxx = P.kxx, best_f = P.best_f, out_sign = P.out_sign;
    const int kind = P.kind, maximize = P.maximize;
    double* ks = dyn;
    double* kd = ks + n;
    double* vv = kd + n;
    const int lane = threadIdx.x;
    const bool want_grad = grad_out != nullptr;
    // the wave's scratch for the adjoint of dlogm (log-Euclidean only): eigenvectors, eigenvalues and their logarithms of the candidate,
    // the gradient with respect to logm(x) as a full matrix, the inner matrix of the Daleckii-Krein formula
    struct FrobLds { double v[D * D], gm[D * D], inner[D * D], lam[D], lg[D]; };
    __shared__ FrobLds sh;
    // ---- candidate features (Mandel order), wave-uniform
    double fx[T];
    double m[T], v[D * D], lam[D], lg[D];
    if constexpr (METRIC == 1) {
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            static_for<r + 1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                double e = xrow[mandel_pos(D, r, c)];
                m[tri(r, c)] = (r == c) ? e : e / kSqrt2;
            });
        });
        sym_eig_reg<D>(m, lam, v);
        static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; lg[k] = log(lam[k]); });
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            static_for<r + 1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                double f = 0.0;
                static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; f = __builtin_fma(v[r * D + k] * lg[k], v[c * D + k], f); });
                fx[mandel_pos(D, r, c)] = (r == c) ? f : f * kSqrt2;
            });
        });
#ifndef GABO_FROB_REGISTER_ADJOINT
        if (want_grad && lane == 0) {          // (every lane holds the same decomposition; the registers are free from here on)
            static_for<D * D>([&](auto ee) { sh.v[decltype(ee)::value] = v[decltype(ee)::value]; });
            static_for<D>([&](auto kk) { sh.lam[decltype(kk)::value] = lam[decltype(kk)::value]; sh.lg[decltype(kk)::value] = lg[decltype(kk)::value]; });
        }
#endif
    } else {
        static_for<T>([&](auto ee) { fx[decltype(ee)::value] = xrow[decltype(ee)::value]; });
    }
    // the reference adds 1e-15 to every MATRIX element of the difference (spd_utils_torch.py:156): Mandel off-diagonals carry sqrt2
    for (int64_t j = lane; j < n; j += 64) {
        double s = 0.0;
        static_for<T>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            const double diff = (fx[e] - Ft[(int64_t)e * n + j]) + (e < D ? 1e-15 : kSqrt2 * 1e-15);
            s = __builtin_fma(diff, diff, s);
        });
        const double kj = exp(-(s * beta));
        ks[j] = os * kj;
        kd[j] = -beta * kj;
    }
    __syncthreads();
    double part = 0.0;
    for (int64_t j = lane; j < n; j += 64) part = __builtin_fma(ks[j], alpha[j], part);
    const double mean = mean0 + wave_sum64(part);
    const double sgn = maximize ? 1.0 : -1.0;
    double g_mean, g_var = 0.0;
    if (kind == GABO_ACQ_POSTERIOR_MEAN) {
        if (lane == 0) *value_out = out_sign * sgn * mean;
        g_mean = sgn;
    } else {
        part = 0.0;
        for (int64_t r = lane; r < n; r += 64) {
            // (linv == linv_t: the symmetric inverse A was handed over instead - see acq_eval)
            vv[r] = strided_dot((sym_inverse ? linv : linv_t) + r, ks, (int)n);      // (L^-1 ks)_r = sum_j L^-T[j][r] ks[j]  (zero for j > r)
            part = sym_inverse ? __builtin_fma(ks[r], vv[r], part) : __builtin_fma(vv[r], vv[r], part);
        }
        const double var = os * kxx - wave_sum64(part);
        const bool clamped = !(var > 1e-9);
        const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
        const double u = sgn * (mean - best_f) / sigma;
        const double pdf = exp(-0.5 * u * u) * 0.3989422804014327;
        const double cdf = 0.5 * (1.0 + erf(u * 0.7071067811865476));
        if (lane == 0) *value_out = out_sign * sigma * (pdf + u * cdf);
        g_mean = sgn * cdf;
        g_var = clamped ? 0.0 : 0.5 * pdf / sigma;
    }
    if (!want_grad) return;
    __syncthreads();
    // ---- G = d(out_sign acq)/d(features) = sum_j w_j 2 (fx - F_j + eps),  w_j = d/d(d_j^2)
    double gacc[T];
    static_for<T>([&](auto ee) { gacc[decltype(ee)::value] = 0.0; });
    for (int64_t j = lane; j < n; j += 64) {
        double ws = 0.0;
        if (kind != GABO_ACQ_POSTERIOR_MEAN) ws = sym_inverse ? vv[j] : strided_dot(linv + j, vv, (int)n);     // (L^-T v)_j = sum_r L^-1[r][j] v[r]  (zero for r < j)
        const double gk = out_sign * os * (g_mean * alpha[j] - 2.0 * g_var * ws);
        const double wj = 2.0 * gk * kd[j];
        static_for<T>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            const double diff = (fx[e] - Ft[(int64_t)e * n + j]) + (e < D ? 1e-15 : kSqrt2 * 1e-15);
            gacc[e] = __builtin_fma(wj, diff, gacc[e]);
        });
    }
    static_for<T>([&](auto ee) { gacc[decltype(ee)::value] = wave_sum64(gacc[decltype(ee)::value]); });
    if constexpr (METRIC == 2) {
        if (lane == 0) static_for<T>([&](auto ee) { grad_out[decltype(ee)::value] = gacc[decltype(ee)::value]; });
    } else {
#ifdef GABO_FROB_REGISTER_ADJOINT      /* rounds 2-5a: the adjoint unrolled in every lane's registers (D^4 products inline, 512-register kernels at D = 7, 8) */
        // adjoint of dlogm at x:  V ((V^T Gm V) o Fdd) V^T,  Fdd_kl = (log l_k - log l_l)/(l_k - l_l), Fdd_kk = 1/l_k
        double inner[D * D];
        static_for<D>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            static_for<D>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                double sacc = 0.0;
                static_for<D>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    double t = 0.0;      // (Gm V)[r][b]
                    static_for<D>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        constexpr int hi = r > c ? r : c, lo = r > c ? c : r;
                        const double gm = (r == c) ? gacc[mandel_pos(D, hi, lo)] : gacc[mandel_pos(D, hi, lo)] / kSqrt2;
                        t = __builtin_fma(gm, v[c * D + b], t);
                    });
                    sacc = __builtin_fma(v[r * D + a], t, sacc);
                });
                const double la = lam[a], lb = lam[b];
                const double meanl = 0.5 * (la + lb), dl = la - lb;
                const double z = dl / (2.0 * meanl), z2 = z * z;
                const double fdd = (__builtin_fabs(z) < 1e-3) ? (1.0 + z2 * (1.0 / 3.0 + z2 * (0.2 + z2 * (1.0 / 7.0)))) / meanl
                                                               : (lg[a] - lg[b]) / dl;
                inner[a * D + b] = sacc * fdd;
            });
        });
        if (lane == 0) {
            static_for<D>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<r + 1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    double f1 = 0.0, f2 = 0.0;       // (r, c) and (c, r): averaged like symmetric_matrix_to_vector_mandel
                    static_for<D>([&](auto aa) {
                        constexpr int a = decltype(aa)::value;
                        static_for<D>([&](auto bb) {
                            constexpr int b = decltype(bb)::value;
                            f1 = __builtin_fma(v[r * D + a] * inner[a * D + b], v[c * D + b], f1);
                            f2 = __builtin_fma(v[c * D + a] * inner[a * D + b], v[r * D + b], f2);
                        });
                    });
                    grad_out[mandel_pos(D, r, c)] = (r == c) ? f1 : 0.5 * (kSqrt2 * f1 + kSqrt2 * f2);
                });
            });
        }
#else
        // adjoint of dlogm at x:  V ((V^T Gm V) o Fdd) V^T,  Fdd_kl = (log l_k - log l_l)/(l_k - l_l), Fdd_kk = 1/l_k - shared by the wave through
        // LDS (round 5): lane a D + b forms entry (a, b) of the inner matrix, lane k entry k of the result.  The same products in the same
        // order as the register form above (which every lane evaluated in full, D^4 of them inline: the source of the 512-register, 4 k-spill
        // instantiations at D = 7, 8 - two of which were wrong as compiled): bit-identical results.
        if (lane == 0) {
            static_for<D>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<D>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    constexpr int hi = r > c ? r : c, lo = r > c ? c : r;
                    sh.gm[r * D + c] = (r == c) ? gacc[mandel_pos(D, hi, lo)] : gacc[mandel_pos(D, hi, lo)] / kSqrt2;
                });
            });
        }
        __syncthreads();
        for (int idx = lane; idx < D * D; idx += 64) {
            const int a = idx / D, b = idx - a * D;
            double sacc = 0.0;
            for (int r = 0; r < D; ++r) {
                double t = 0.0;      // (Gm V)[r][b]
                for (int c = 0; c < D; ++c) t = __builtin_fma(sh.gm[r * D + c], sh.v[c * D + b], t);
                sacc = __builtin_fma(sh.v[r * D + a], t, sacc);
            }
            const double la = sh.lam[a], lb = sh.lam[b];
            const double meanl = 0.5 * (la + lb), dl = la - lb;
            const double z = dl / (2.0 * meanl), z2 = z * z;
            const double fdd = (__builtin_fabs(z) < 1e-3) ? (1.0 + z2 * (1.0 / 3.0 + z2 * (0.2 + z2 * (1.0 / 7.0)))) / meanl
                                                           : (sh.lg[a] - sh.lg[b]) / dl;
            sh.inner[idx] = sacc * fdd;
        }
        __syncthreads();
        for (int k = lane; k < T; k += 64) {
            int r = 0;
            while (tri(r + 1, 0) <= k) ++r;
            const int c = k - tri(r, 0);
            double f1 = 0.0, f2 = 0.0;       // (r, c) and (c, r): averaged like symmetric_matrix_to_vector_mandel
            for (int a = 0; a < D; ++a) {
                for (int b = 0; b < D; ++b) {
                    f1 = __builtin_fma(sh.v[r * D + a] * sh.inner[a * D + b], sh.v[c * D + b], f1);
                    f2 = __builtin_fma(sh.v[c * D + a] * sh.inner[a * D + b], sh.v[r * D + b], f2);
                }
            }
            grad_out[mandel_pos(D, r, c)] = (r == c) ? f1 : 0.5 * (kSqrt2 * f1 + kSqrt2 * f2);
        }
        __syncthreads();         // (the scratch is reused by the next evaluation of this wave)
#endif
    }
}

// metric dispatch: GABO_METRIC_* bits of P.flags are resolved by the host into the METRIC template parameter
template <int D, int METRIC>
__device__ __forceinline__ void acq_eval_any(const double* __restrict__ xrow, const AcqParams& P, double* __restrict__ value_out,
                                             double* __restrict__ grad_out, double* __restrict__ F, AcqLds<D>& L, double* dyn,
                                             int* __restrict__ status, int64_t index) {
    if constexpr (METRIC == 0) acq_eval<D>(xrow, P, value_out, grad_out, F, L, dyn, status, index);
    else acq_eval_frob<D, METRIC>(xrow, P, value_out, grad_out, dyn);
}

}  // namespace gabo
