// Device-side bodies of the SPD truncated-CG state machine (see spd_tcg.hip): shared by the stand-alone kernels and by the
// trust-region iteration kernel (spd_tr.hip).  One wave (64-thread block) per restart.
#pragma once
#include "gabo_device.hpp"
#include "lds_linalg.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

enum { TCG_NEGATIVE_CURVATURE = 0, TCG_EXCEEDED_TR, TCG_REACHED_TARGET_LINEAR, TCG_REACHED_TARGET_SUPERLINEAR, TCG_MAX_INNER_ITER,
       TCG_MODEL_INCREASED, TCG_REACHED_CONSTRAINTS };
enum { SC_DELTA = 0, SC_E_PE, SC_E_PD, SC_D_PD, SC_Z_R, SC_MODEL, SC_NORM_R0, SC_C_FD, SC_COUNT };
constexpr int kMaxCons = 8;
#ifndef GABO_FD_EPS
#define GABO_FD_EPS 6.103515625e-05    /* "how far do we look": 2^-14, approximate_hessian.py:40 (rounds 2-4 had 2^-13 here; A/B: tools/ab_build.py) */
#endif

struct TcgWs {
    double *chol, *g_w, *eta_w, *heta_w, *r_w, *delta_w, *expm, *w_ones, *scal, *gc_w, *fc, *fcg_pe;
    int *stop, *running, *counters;
    size_t bytes;
    int64_t index_base;      // added to the restart index when an error is reported (a block-private workspace is indexed with 0)
};

static __host__ __device__ inline TcgWs tcg_layout(void* base, int64_t R, int d, int C) {
    TcgWs w;
    double* p = (double*)base;
    const int64_t m = R * d * d;
    w.chol = p;     p += m;
    w.g_w = p;      p += m;
    w.eta_w = p;    p += m;
    w.heta_w = p;   p += m;
    w.r_w = p;      p += m;
    w.delta_w = p;  p += m;
    w.expm = p;     p += m;
    w.w_ones = p;   p += R * d;
    w.scal = p;     p += R * SC_COUNT;
    w.gc_w = p;     p += (int64_t)C * m;
    w.fc = p;       p += R * (C > 0 ? C : 1);
    w.fcg_pe = p;   p += R * (C > 0 ? C : 1);
    int* q = (int*)p;
    w.stop = q;     q += R;
    w.running = q;  q += R;
    w.counters = q; q += 4;
    w.bytes = (size_t)((char*)q - (char*)base);
    w.index_base = 0;
    return w;
}

static __device__ __forceinline__ double wave_sum(double v) { return wave_allsum(v); }      // gabo_device.hpp

static __device__ __forceinline__ double wave_dot(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += 64) s = __builtin_fma(a[e], b[e], s);
    return wave_sum(s);
}

// z = precon(r): the reference adds 1e-30 to every element of a direction whose elements sum to exactly zero
// (manifold_optimize.py:190-193); in whitened coordinates that is + 1e-30 (L^-1 1)(L^-1 1)^T.
static __device__ __forceinline__ double precon_entry(double r, bool zero_sum, const double* w1, int e, int d) {
    return zero_sum ? r + 1e-30 * w1[e / d] * w1[e % d] : r;
}

// sum of the elements of the UNwhitened matrix L r~ L^T = (L^T 1)^T r~ (L^T 1)
static __device__ double unwhitened_sum(const double* L, const double* rw, int d) {
    double s = 0.0;
    for (int e = threadIdx.x; e < d * d; e += 64) {
        int a = e / d, b = e - a * d;
        double ca = 0.0, cb = 0.0;      // column sums of L
        for (int k = a; k < d; ++k) ca += L[k * d + a];
        for (int k = b; k < d; ++k) cb += L[k * d + b];
        s = __builtin_fma(ca * rw[e], cb, s);
    }
    return wave_sum(s);
}

// x, g: this restart's d x d matrices; gc: constraint gradients (C x R x d x d), fc: constraint values (R x C).
// lds: 5 d^2 doubles.  Every thread of the (one-wave) block calls it.
static __device__ void tcg_begin(const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ gc,
                                 const double* __restrict__ fc, bool active, double delta_tr, const TcgWs& w, int64_t i, int64_t R,
                                 int d, int C, int* __restrict__ status, double* lds, bool reuse = false) {
    const int dd = d * d;
    double* M0 = lds;          // L
    double* M1 = M0 + dd;      // W = L^-1
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* M4 = M3 + dd;
    double* gw = w.g_w + i * dd;
    double* rw = w.r_w + i * dd;
    if (reuse) {
        // x and g are those of the previous call for this restart (its proposal was rejected): the factor, the whitened gradient,
        // L^-1 1 and the whitened constraint gradients in the workspace are still valid - only the iteration state is reset.  M0 / M3
        // are refilled with the stored values, so everything below sees bit for bit what the full path computes.
        lds_load(w.chol + i * dd, M0, d);
        lds_load(gw, M3, d);
        for (int e = threadIdx.x; e < dd; e += 64) {
            rw[e] = M3[e];
            w.eta_w[i * dd + e] = 0.0;
            w.heta_w[i * dd + e] = 0.0;
        }
        __syncthreads();
    } else {
    lds_load(x, M0, d);
    lds_symmetrize(M0, M4, d);
    bool ok = lds_cholesky(M0, d);
    lds_tri_inverse(M0, M1, d);
    if (!ok && threadIdx.x == 0 && status) {
        if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)(i + w.index_base);
    }
    lds_load(g, M2, d);
    lds_symmetrize(M2, M4, d);
    lds_congruence(M1, M2, M3, M4, d);      // g~ = W g W^T
    lds_symmetrize(M3, M4, d);
    for (int e = threadIdx.x; e < dd; e += 64) {
        w.chol[i * dd + e] = M0[e];
        gw[e] = M3[e];
        rw[e] = M3[e];
        w.eta_w[i * dd + e] = 0.0;
        w.heta_w[i * dd + e] = 0.0;
    }
    for (int r = threadIdx.x; r < d; r += 64) {
        double s = 0.0;
        for (int k = 0; k <= r; ++k) s += M1[r * d + k];
        w.w_ones[i * d + r] = s;
    }
    __syncthreads();
    }
    const double rr = wave_dot(M3, M3, dd);
    const double usum = unwhitened_sum(M0, M3, d);
    const bool zero_sum = usum == 0.0;
    double zr = 0.0;
    for (int e = threadIdx.x; e < dd; e += 64) {
        double z = precon_entry(M3[e], zero_sum, w.w_ones + i * d, e, d);
        w.delta_w[i * dd + e] = -z;
        zr = __builtin_fma(z, M3[e], zr);
    }
    zr = wave_sum(zr);
    for (int k = 0; gc != nullptr && !reuse && k < C; ++k) {
        lds_load(gc + ((int64_t)k * R + i) * dd, M2, d);
        lds_symmetrize(M2, M4, d);
        lds_congruence(M1, M2, M3, M4, d);
        lds_symmetrize(M3, M4, d);
        for (int e = threadIdx.x; e < dd; e += 64) w.gc_w[((int64_t)k * R + i) * dd + e] = M3[e];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* sc = w.scal + i * SC_COUNT;
        sc[SC_DELTA] = delta_tr;
        sc[SC_E_PE] = 0.0;
        sc[SC_E_PD] = 0.0;
        sc[SC_D_PD] = zr;
        sc[SC_Z_R] = zr;
        sc[SC_MODEL] = 0.0;
        sc[SC_NORM_R0] = __builtin_sqrt(rr > 0.0 ? rr : 0.0);
        if (!reuse) sc[SC_C_FD] = 0.0;        // (reuse: the first FD point's c may be kept together with its E, see tr_propose_body)
        w.stop[i] = TCG_MAX_INNER_ITER;
        w.running[i] = active ? 1 : 0;
        for (int k = 0; k < C; ++k) { if (fc != nullptr) w.fc[i * C + k] = fc[i * C + k]; w.fcg_pe[i * C + k] = 0.0; }
    }
}

// tcg_begin for D <= 8 in registers (round 6): the same quantities - L = chol(x), W = L^-1, the whitened gradient g~ = W sym(g) W^T, L^-1 1, the
// preconditioned first direction and the scalars of the iteration - computed by EVERY lane redundantly from wave-uniform operands with compile-time
// indices, then written to the workspace by the lanes that own an entry.  The LDS-phased form above is ~20 barrier-separated steps on 5 x 5 tiles
// (14 k of an accepted iteration's 145 k cycles at d = 5, tools/tr_clocks.py); here the chain is the Cholesky recurrence and two small products.
// Host-evaluated constraint gradients (gc != null: the multi-launch plans with opaque callables) are whitened by the LDS loop of tcg_begin, for
// which W is left in lds[D^2 ...].  Not for `reuse` (the caller takes tcg_begin then: a copy, no arithmetic).  lds: 5 D^2 doubles.
template <int D>
static __device__ __forceinline__ void tcg_begin_reg(const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ gc,
                                                     const double* __restrict__ fc, bool active, double delta_tr, const TcgWs& w, int64_t i,
                                                     int64_t R, int C, int* __restrict__ status, double* lds) {
    constexpr int T = tri_size(D);
    constexpr int dd = D * D;
    const int lane = threadIdx.x;
    double a[T], wv[T], gs[T];
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            a[tri(r, c)] = 0.5 * (x[r * D + c] + x[c * D + r]);
            gs[tri(r, c)] = 0.5 * (g[r * D + c] + g[c * D + r]);
        });
    });
    bool bad = false;
    static_for<D>([&](auto cc) {          // Cholesky (the recurrence of mandel_cholesky, spd_prep.hpp)
        constexpr int c = decltype(cc)::value;
        double piv = a[tri(c, c)];
        static_for<c>([&](auto kk) { constexpr int k = decltype(kk)::value; piv = __builtin_fma(-a[tri(c, k)], a[tri(c, k)], piv); });
        if (!(piv > 0.0)) bad = true;
        const double inv = rsqrt_nz(piv);
        a[tri(c, c)] = piv * inv;
        static_for<D - c - 1>([&](auto rr) {
            constexpr int r = c + 1 + decltype(rr)::value;
            double sacc = a[tri(r, c)];
            static_for<c>([&](auto kk) { constexpr int k = decltype(kk)::value; sacc = __builtin_fma(-a[tri(r, k)], a[tri(c, k)], sacc); });
            a[tri(r, c)] = sacc * inv;
        });
    });
    if (bad && lane == 0 && status) {
        if (atomicCAS(status, 0, GABO_ERR_NOT_SPD) == 0) status[1] = (int)(i + w.index_base);
    }
    static_for<D>([&](auto cc) { constexpr int c = decltype(cc)::value; wv[tri(c, c)] = rcp(a[tri(c, c)]); });
    static_for<D>([&](auto cc) {          // W = L^-1 (lower_inverse, spd_prep.hpp)
        constexpr int c = decltype(cc)::value;
        static_for<D - c - 1>([&](auto rr) {
            constexpr int r = c + 1 + decltype(rr)::value;
            double sacc = 0.0;
            static_for<r - c>([&](auto kk) { constexpr int k = c + decltype(kk)::value; sacc = __builtin_fma(a[tri(r, k)], wv[tri(k, c)], sacc); });
            wv[tri(r, c)] = -sacc * wv[tri(r, r)];
        });
    });
    // g~ = W gs W^T, lower triangle: t = W gs (full rows), g~[r][c] = sum_{k <= c} t[r][k] W[c][k]
    double gw[T];
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        double t[D];
        static_for<D>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            double sacc = 0.0;
            static_for<r + 1>([&](auto mm) {
                constexpr int m = decltype(mm)::value;
                constexpr int hi = m > k ? m : k, lo = m > k ? k : m;
                sacc = __builtin_fma(wv[tri(r, m)], gs[tri(hi, lo)], sacc);
            });
            t[k] = sacc;
        });
        static_for<r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            double sacc = 0.0;
            static_for<c + 1>([&](auto kk) { constexpr int k = decltype(kk)::value; sacc = __builtin_fma(t[k], wv[tri(c, k)], sacc); });
            gw[tri(r, c)] = sacc;
        });
    });
    // <g~, g~>, the sum of the elements of the unwhitened L g~ L^T = (L^T 1)^T g~ (L^T 1), L^-1 1
    double rr2 = 0.0, cs[D], ones[D];
    static_for<D>([&](auto aa) {
        constexpr int c = decltype(aa)::value;
        double sacc = 0.0;
        static_for<D - c>([&](auto kk) { constexpr int k = c + decltype(kk)::value; sacc += a[tri(k, c)]; });
        cs[c] = sacc;
        double o = 0.0;
        static_for<c + 1>([&](auto kk) { o += wv[tri(c, decltype(kk)::value)]; });
        ones[c] = o;
    });
    double usum = 0.0;
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<D>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int hi = r > c ? r : c, lo = r > c ? c : r;
            const double v = gw[tri(hi, lo)];
            rr2 = __builtin_fma(v, v, rr2);
            usum = __builtin_fma(cs[r] * v, cs[c], usum);
        });
    });
    const bool zero_sum = usum == 0.0;
    double zr = 0.0;
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<D>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int hi = r > c ? r : c, lo = r > c ? c : r;
            const double v = gw[tri(hi, lo)];
            const double z = zero_sum ? v + 1e-30 * ones[r] * ones[c] : v;
            zr = __builtin_fma(z, v, zr);
            if (lane == r * D + c) {          // the lane that owns entry (r, c) writes it
                w.chol[i * dd + r * D + c] = r >= c ? a[tri(hi, lo)] : 0.0;
                w.g_w[i * dd + r * D + c] = v;
                w.r_w[i * dd + r * D + c] = v;
                w.eta_w[i * dd + r * D + c] = 0.0;
                w.heta_w[i * dd + r * D + c] = 0.0;
                w.delta_w[i * dd + r * D + c] = -z;
                if (gc != nullptr) lds[dd + r * D + c] = r >= c ? wv[tri(hi, lo)] : 0.0;
            }
        });
        if (lane == r) w.w_ones[i * D + r] = ones[r];
    });
    static_assert(dd <= 64, "one lane per matrix entry (D <= 8)");
    __syncthreads();
    for (int k = 0; gc != nullptr && k < C; ++k) {          // host-evaluated constraint gradients: the LDS loop of tcg_begin
        double* M1 = lds + dd;
        double* M2 = M1 + dd;
        double* M3 = M2 + dd;
        double* M4 = M3 + dd;
        lds_load(gc + ((int64_t)k * R + i) * dd, M2, D);
        lds_symmetrize(M2, M4, D);
        lds_congruence(M1, M2, M3, M4, D);
        lds_symmetrize(M3, M4, D);
        for (int e = lane; e < dd; e += 64) w.gc_w[((int64_t)k * R + i) * dd + e] = M3[e];
        __syncthreads();
    }
    if (lane == 0) {
        double* sc = w.scal + i * SC_COUNT;
        sc[SC_DELTA] = delta_tr;
        sc[SC_E_PE] = 0.0;
        sc[SC_E_PD] = 0.0;
        sc[SC_D_PD] = zr;
        sc[SC_Z_R] = zr;
        sc[SC_MODEL] = 0.0;
        sc[SC_NORM_R0] = __builtin_sqrt(rr2 > 0.0 ? rr2 : 0.0);
        sc[SC_C_FD] = 0.0;
        w.stop[i] = TCG_MAX_INNER_ITER;
        w.running[i] = active ? 1 : 0;
        for (int k = 0; k < C; ++k) { if (fc != nullptr) w.fc[i * C + k] = fc[i * C + k]; w.fcg_pe[i * C + k] = 0.0; }
    }
}

// use_rand (robust_trust_regions.py:176-181, 407-452): tCG starts from a tiny random tangent vector eta0 instead of zero, with
// Heta0 = hess(x, eta0) given by the caller, r = g + Heta0, no preconditioner (z = r), delta = -r, e_Pe = <eta0, eta0>,
// e_Pd = <eta0, delta>, model = <eta0, g> + <eta0, Heta0> / 2.  Called after tcg_begin for the same restart (which left the factor,
// the whitened gradient and the constraints' state in the workspace); eta0, heta0: this restart's d x d tangent vectors at x.
// lds: 5 d^2 doubles.
static __device__ void tcg_begin_rand(const TcgWs& w, int64_t i, int64_t R, int d, int C, const double* __restrict__ eta0,
                                      const double* __restrict__ heta0, double* lds) {
    const int dd = d * d;
    double* M0 = lds;          // L
    double* M1 = M0 + dd;      // W = L^-1
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;      // eta~
    double* M4 = M3 + dd;
    lds_load(w.chol + i * dd, M0, d);
    lds_tri_inverse(M0, M1, d);
    lds_load(heta0, M2, d);
    lds_symmetrize(M2, M4, d);
    lds_congruence(M1, M2, M3, M4, d);      // Heta~ = W Heta0 W^T
    lds_symmetrize(M3, M4, d);
    const double* gw = w.g_w + i * dd;
    double* rw = w.r_w + i * dd;
    double rr = 0.0;
    for (int e = threadIdx.x; e < dd; e += 64) {
        const double h = M3[e], r = gw[e] + h;
        w.heta_w[i * dd + e] = h;
        M0[e] = h;                          // (L is not needed any more)
        rw[e] = r;
        w.delta_w[i * dd + e] = -r;
        rr = __builtin_fma(r, r, rr);
    }
    rr = wave_sum(rr);
    __syncthreads();
    lds_load(eta0, M2, d);
    lds_symmetrize(M2, M4, d);
    lds_congruence(M1, M2, M3, M4, d);      // eta~ = W eta0 W^T
    lds_symmetrize(M3, M4, d);
    double ee = 0.0, er = 0.0, eg = 0.0, eh = 0.0;
    for (int e = threadIdx.x; e < dd; e += 64) {
        const double a = M3[e];
        w.eta_w[i * dd + e] = a;
        ee = __builtin_fma(a, a, ee);
        er = __builtin_fma(a, rw[e], er);
        eg = __builtin_fma(a, gw[e], eg);
        eh = __builtin_fma(a, M0[e], eh);
    }
    ee = wave_sum(ee);
    er = wave_sum(er);
    eg = wave_sum(eg);
    eh = wave_sum(eh);
    // the linearised constraints start from <grad c_k, eta0>, not from zero (constrained_trust_regions.py:512-516: `fcgradx_Pe[c] = inner(x,
    // fgradx_cons[c], eta)`) - of the order of Delta_cons = 1e-6 for the 1e-6 start
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        const double ge = wave_dot(w.gc_w + ((int64_t)k * R + i) * dd, M3, dd);
        if (threadIdx.x == 0) w.fcg_pe[i * C + k] = ge;
    }
    if (threadIdx.x == 0) {
        double* sc = w.scal + i * SC_COUNT;
        sc[SC_E_PE] = ee;
        sc[SC_E_PD] = -er;
        sc[SC_D_PD] = rr;
        sc[SC_Z_R] = rr;
        sc[SC_MODEL] = eg + 0.5 * eh;
        sc[SC_NORM_R0] = __builtin_sqrt(rr > 0.0 ? rr : 0.0);
    }
}

// FD point of get_hessianfd (approximate_hessian.py:30-47): c = 2^-14 / ||delta||_x, x1 = retr(x, c delta) = L expm(c delta~) L^T.
// Restarts that are not running get x1 = x (their gradient is evaluated but not used).
// x_fd: this restart's Mandel row (global or LDS).  lds: 4 d^2 + 2 doubles.
static __device__ void tcg_fd_point(const TcgWs& w, int64_t i, int d, double* __restrict__ x_fd, double* lds) {
    const int dd = d * d, dv = d * (d + 1) / 2;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* cs = M3 + dd;
    const bool run = w.running[i] != 0;
    lds_load(w.chol + i * dd, M0, d);
    double c = 0.0;
    bool tiny = true;
    if (run) {
        const double* dl = w.delta_w + i * dd;
        const double nrm = __builtin_sqrt(wave_dot(dl, dl, dd));
        tiny = nrm < 1e-15;
        c = GABO_FD_EPS / (tiny ? 1.0 : nrm);
        for (int e = threadIdx.x; e < dd; e += 64) M1[e] = c * dl[e];
        __syncthreads();
        // E = expm(A), ||A||_F = 2^-14 by construction: I + A + A^2/2 + A^3/6 leaves ||A||^4/24 < 6e-19 (the eigen-decomposition
        // route costs ~240 barrier phases for the same matrix and is less accurate relative to E - I, which is what matters here)
        lds_mm(M1, M1, M2, d, false, false);                 // A^2
        lds_mm(M2, M1, M3, d, false, false);                 // A^3
        for (int e = threadIdx.x; e < dd; e += 64) {
            const double a1 = M1[e], a2 = M2[e], a3 = M3[e];
            M3[e] = ((e / d == e % d) ? 1.0 : 0.0) + (a1 + (0.5 * a2 + a3 / 6.0));
        }
        __syncthreads();
        lds_symmetrize(M3, M2, d);                           // E
    } else {
        for (int e = threadIdx.x; e < dd; e += 64) M3[e] = (e / d == e % d) ? 1.0 : 0.0;
        __syncthreads();
    }
    for (int e = threadIdx.x; e < dd; e += 64) w.expm[i * dd + e] = M3[e];
    if (threadIdx.x == 0) w.scal[i * SC_COUNT + SC_C_FD] = tiny ? -c : c;      // sign bit carries the "tiny" flag
    lds_congruence(M0, M3, M1, M2, d);                       // x1 = L E L^T
    for (int e = threadIdx.x; e < dv; e += 64) {
        int k = 0;
        while (k + 1 < d && (k + 1) * d - (k + 1) * k / 2 <= e) ++k;
        int cc = e - (k * d - k * (k - 1) / 2);
        int r = cc + k;
        x_fd[e] = (k == 0) ? M1[r * d + cc] : 0.5 * (kSqrt2 * M1[r * d + cc] + kSqrt2 * M1[cc * d + r]);
    }
}

struct ConsStep { double cin, tau; };

// violation of the linearised constraints after `step` along delta and the step that stops at Delta_cons
// (constrained_trust_regions.py:583-640, same algebra as batched_trust_regions._tcg_step.cons_step)
static __device__ ConsStep cons_step(double step, const double* fc, const double* fcg_pe, const double* fcg_pd, int C, int neq,
                                     double dc2) {
    double cin = 0.0, qa = 0.0, qb1 = 0.0, qb2 = 0.0, qc1 = 0.0, qc2 = 0.0, qc3 = 0.0;
    for (int k = 0; k < C; ++k) {
        double term = fc[k] + fcg_pe[k] + step * fcg_pd[k];
        const bool ineq = k >= neq;
        if (ineq && term > 0.0) term = 0.0;
        cin += term * term;
        const double m = (!ineq || term < 0.0) ? 1.0 : 0.0;
        qa += m * fcg_pd[k] * fcg_pd[k];
        qb1 += m * fc[k] * fcg_pd[k];
        qb2 += m * fcg_pe[k] * fcg_pd[k];
        qc1 += m * fc[k] * fc[k];
        qc2 += m * fc[k] * fcg_pe[k];
        qc3 += m * fcg_pe[k] * fcg_pe[k];
    }
    const double qb = 2.0 * (qb1 + qb2);
    const double qc = qc1 + 2.0 * qc2 + qc3 - dc2;
    const double disc = qb * qb - 4.0 * qa * qc;
    ConsStep r;
    r.cin = cin;
    r.tau = disc >= 0.0 ? (-qb + __builtin_sqrt(disc)) / (2.0 * qa) : 0.0;
    return r;
}

// Per-restart views of the tCG state (vectors of length L = d^2 whitened matrices for S^d_++, or the ambient dimension for the sphere)
struct TcgVecs {
    const double* g;            // gradient (whitened for SPD)
    double *eta, *heta, *r, *delta;
    const double* gc;           // constraint gradients: gc + k * gc_stride, k < C
    int64_t gc_stride;
    double* scal;               // SC_COUNT scalars
    const double* fc;           // C constraint values
    double* fcg_pe;             // C
    int *stop, *running;
};

// The manifold-independent part of one tCG iteration (robust_trust_regions.py:476-568, constrained_trust_regions.py:530-732): Hd and a
// copy dl of delta are in LDS (length L), the inner product is the plain dot of the stored vectors.  s0..s3: four LDS scratch vectors
// of length L (may alias whatever produced Hd).  Precon: zero_sum(rn) -> bool, entry(r_e, zero_sum, e) -> z_e  (the reference's
// "+1e-30 when the elements sum to zero" preconditioner in the caller's coordinates).
template <class Precon>
static __device__ bool tcg_step_core(const TcgVecs& v, int L, int C, double* Hd, double* dl, double* s0, double* s1, double* s2, int neq,
                                     double delta_cons, double theta, double kappa, int mininner, int iter, Precon precon) {
    double* sc = v.scal;
    const double* gw = v.g;
    double* eta = v.eta;
    double* heta = v.heta;
    double* rw = v.r;
    const double Delta = sc[SC_DELTA], e_Pe = sc[SC_E_PE], e_Pd = sc[SC_E_PD], d_Pd = sc[SC_D_PD], z_r = sc[SC_Z_R];
    const double dc2 = delta_cons * delta_cons;
    const double d_Hd = wave_dot(dl, Hd, L);
    const bool nz = d_Hd != 0.0;
    const double alpha = nz ? z_r / d_Hd : 0.0;
    const double e_Pe_new = nz ? e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd : e_Pe;
    const double Delta2 = Delta * Delta;
    double fcl[kMaxCons], fpe[kMaxCons], fpd[kMaxCons];
    for (int k = 0; k < C; ++k) {
        fcl[k] = v.fc[k];
        fpe[k] = v.fcg_pe[k];
        fpd[k] = wave_dot(v.gc + (int64_t)k * v.gc_stride, dl, L);
    }
    int stop = -1;
    double step = 0.0;       // eta += step * delta, Heta += step * Hd when leaving
    // ---- leave through the trust-region boundary / negative curvature
    if (d_Hd <= 0.0 || e_Pe_new >= Delta2) {
        double tau = (-e_Pd + __builtin_sqrt(e_Pd * e_Pd + d_Pd * (Delta2 - e_Pe))) / d_Pd;
        stop = d_Hd <= 0.0 ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;
        if (C > 0) {
            if (tau != tau) tau = 0.0;
            ConsStep cst = cons_step(tau, fcl, fpe, fpd, C, neq, dc2);
            if (cst.cin > dc2) {
                tau = cst.tau;
                if (d_Hd > 0.0) stop = TCG_REACHED_CONSTRAINTS;
            }
        }
        step = tau;
    } else if (C > 0) {
        // ---- leave because the linearised constraints are reached inside the trust region
        ConsStep cst = cons_step(alpha, fcl, fpe, fpd, C, neq, dc2);
        if (cst.cin > dc2) { stop = TCG_REACHED_CONSTRAINTS; step = cst.tau; }
    }
    if (stop >= 0) {
        for (int e = threadIdx.x; e < L; e += 64) {
            eta[e] = eta[e] + step * dl[e];
            heta[e] = heta[e] + step * Hd[e];
        }
        if (threadIdx.x == 0) { *v.stop = stop; *v.running = 0; }
        return false;
    }
    // ---- tentative step; reject it if the model did not decrease
    double* ne = s0;
    double* nh = s1;
    double m1 = 0.0, m2 = 0.0;
    for (int e = threadIdx.x; e < L; e += 64) {
        double a = eta[e] + alpha * dl[e], b = heta[e] + alpha * Hd[e];
        ne[e] = a;
        nh[e] = b;
        m1 = __builtin_fma(a, gw[e], m1);
        m2 = __builtin_fma(a, b, m2);
    }
    const double new_model = wave_sum(m1) + 0.5 * wave_sum(m2);
    if (!(new_model < sc[SC_MODEL])) {
        if (threadIdx.x == 0) { *v.stop = TCG_MODEL_INCREASED; *v.running = 0; }
        return false;
    }
    double rr = 0.0;
    double* rn = s2;
    for (int e = threadIdx.x; e < L; e += 64) {
        eta[e] = ne[e];
        heta[e] = nh[e];
        double r = rw[e] + alpha * Hd[e];
        rw[e] = r;
        rn[e] = r;
        rr = __builtin_fma(r, r, rr);
    }
    rr = wave_sum(rr);
    __syncthreads();
    const double norm_r = __builtin_sqrt(rr > 0.0 ? rr : 0.0);
    bool running = true;
    // ---- residual small enough
    if (iter >= mininner) {
        const double nr0 = sc[SC_NORM_R0];
        // (theta = 1 - pymanopt's default, the reference's setting - needs no pow: ~300 dependent instructions of every tCG step)
        const double p = theta == 1.0 ? nr0 : pow(nr0, theta);
        const double target = nr0 * (p < kappa ? p : kappa);
        if (norm_r <= target) {
            stop = kappa < p ? TCG_REACHED_TARGET_LINEAR : TCG_REACHED_TARGET_SUPERLINEAR;
            running = false;
        }
    }
    if (threadIdx.x == 0) {
        sc[SC_MODEL] = new_model;
        sc[SC_E_PE] = e_Pe_new;
        if (!running) { *v.stop = stop; *v.running = 0; }
    }
    if (!running) return false;
    // ---- next search direction
    const bool zero_sum = precon.zero_sum(rn);
    double zr = 0.0;
    double* zv = s1;
    for (int e = threadIdx.x; e < L; e += 64) {
        double z = precon.entry(rn[e], zero_sum, e);
        zv[e] = z;
        zr = __builtin_fma(z, rn[e], zr);
    }
    const double z_r_new = wave_sum(zr);
    const double beta = z_r_new / z_r;
    for (int e = threadIdx.x; e < L; e += 64) v.delta[e] = -zv[e] + beta * dl[e];
    if (threadIdx.x == 0) {
        sc[SC_E_PD] = beta * (e_Pd + alpha * d_Pd);
        sc[SC_D_PD] = z_r_new + beta * beta * d_Pd;
        sc[SC_Z_R] = z_r_new;
        for (int k = 0; k < C; ++k) v.fcg_pe[k] = fpe[k] + alpha * fpd[k];
    }
    return true;
}

// the preconditioner of manifold_optimize.py:190-193 in whitened SPD coordinates (see precon_entry / unwhitened_sum)
struct SpdPrecon {
    double* Lbuf;           // LDS, d x d: receives chol(x)
    const double* chol;     // global
    const double* w1;       // row sums of L^-1
    int d;
    bool off;               // use_rand: "and therefore, no preconditioner" (robust_trust_regions.py:411, 424-427): z = r
    __device__ bool zero_sum(const double* rn) const {
        if (off) return false;
        lds_load(chol, Lbuf, d);
        return unwhitened_sum(Lbuf, rn, d) == 0.0;
    }
    __device__ double entry(double r, bool zs, int e) const { return precon_entry(r, zs, w1, e, d); }
};

// egrad_fd: this restart's Euclidean gradient at the FD point (Mandel row, global or LDS).  lds: 5 d^2 doubles.
// Returns true while the restart keeps running.  `iter` = index of this inner iteration (0-based).
static __device__ bool tcg_step(const TcgWs& w, int64_t i, int64_t R, int d, int C, const double* __restrict__ egrad_fd, int neq,
                                double delta_cons, double theta, double kappa, int mininner, int iter, double* lds, bool no_precon = false) {
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* M4 = M3 + dd;
    if (w.running[i] == 0) return false;
    double* sc = w.scal + i * SC_COUNT;
    // ---- Hessian-vector product by finite differences, whitened:  Hd~ = (E L^T sym(eg1) L E - g~) / c
    lds_load(w.chol + i * dd, M0, d);
    lds_load(w.expm + i * dd, M1, d);
    lds_from_mandel(egrad_fd, M2, d);
    lds_mm(M0, M2, M3, d, true, false);     // L^T S
    lds_mm(M3, M0, M4, d, false, false);    // L^T S L
    lds_mm(M1, M4, M3, d, false, false);    // E .
    lds_mm(M3, M1, M4, d, false, false);    // E . E   = g1~
    lds_symmetrize(M4, M3, d);
    const double cs = sc[SC_C_FD];
    const bool tiny = cs <= 0.0;
    const double c = __builtin_fabs(cs);
    const double* gw = w.g_w + i * dd;
    double* Hd = M4;
    double* dl = M2;
    for (int e = threadIdx.x; e < dd; e += 64) {
        Hd[e] = tiny ? 0.0 : Hd[e] / c - gw[e] / c;
        dl[e] = w.delta_w[i * dd + e];
    }
    __syncthreads();
    TcgVecs v{gw, w.eta_w + i * dd, w.heta_w + i * dd, w.r_w + i * dd, w.delta_w + i * dd, w.gc_w + i * dd, (int64_t)R * dd, sc,
              w.fc + i * C, w.fcg_pe + i * C, w.stop + i, w.running + i};
    SpdPrecon pc{M0, w.chol + i * dd, w.w_ones + i * d, d, no_precon};
    return tcg_step_core(v, dd, C, Hd, dl, M0, M1, M3, neq, delta_cons, theta, kappa, mininner, iter, pc);
}

// eta = L eta~ L^T, Heta = L Heta~ L^T, stop reasons
static __device__ void tcg_end(const TcgWs& w, int64_t i, int d, double* __restrict__ eta, double* __restrict__ heta,
                               double* lds) {
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    lds_load(w.chol + i * dd, M0, d);
    lds_load(w.eta_w + i * dd, M1, d);
    lds_congruence(M0, M1, M2, M3, d);
    lds_symmetrize(M2, M3, d);
    lds_store(M2, eta, d);
    __syncthreads();
    lds_load(w.heta_w + i * dd, M1, d);
    lds_congruence(M0, M1, M2, M3, d);
    lds_symmetrize(M2, M3, d);
    lds_store(M2, heta, d);
}

}  // namespace gabo
