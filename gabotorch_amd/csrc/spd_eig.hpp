// Per-lane symmetric eigenvalue solver for small fixed D (register resident, no cross-lane traffic):
//   Householder tridiagonalisation of a packed lower triangle, then square-root-free implicit QL
//   (Pal-Walker-Kahan recurrence, the scheme LAPACK's dsterf uses) on (diag, offdiag^2).
// One lane owns one matrix; all indices into the register arrays are compile-time constants.
//
// This is the arithmetic that replaces the reference's per-pair `torch.symeig` call
// (Riemannian_utils/spd_utils_torch.py:109-110).
#pragma once
#include "gabo_device.hpp"

namespace gabo {

// m: packed lower triangle of a symmetric D x D matrix (destroyed).  Out: dg[0..D-1] diagonal and
// e2[0..D-2] squared off-diagonals of the similar tridiagonal matrix (e2[D-1] = 0).
template <int D>
__device__ __forceinline__ void tridiagonalize(double (&m)[tri_size(D)], double (&dg)[D], double (&e2)[D]) {
    static_for<D - 2>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int n = D - k - 1;  // order of the trailing block; column below the diagonal is x_0..x_{n-1}
        // u = x + sign(x0)|x| e0 ;  H = I - u u^T / hh,  hh = |x|^2 + |x0||x|
        double alpha = m[tri(k + 1, k)];
        double sig = 0.0;
        static_for<n - 1>([&](auto t) { double x = m[tri(k + 2 + decltype(t)::value, k)]; sig = __builtin_fma(x, x, sig); });
        double nn = __builtin_fma(alpha, alpha, sig);
#ifdef GABO_TRIDIAG_ZERO_GUARDS       /* A/B: the round-1 form (two compares and four selects per column for the all-zero column) */
        double nrm = sqrt_pos(nn);
        double u[n];
        u[0] = alpha + copysign_d(nrm, alpha);
        static_for<n - 1>([&](auto t) { u[decltype(t)::value + 1] = m[tri(k + 2 + decltype(t)::value, k)]; });
        double hh = __builtin_fma(__builtin_fabs(alpha), nrm, nn);
        double inv_hh = hh == 0.0 ? 0.0 : rcp(hh);
#else
        // A column that is already zero needs no special case when |x| is floored at 1e-145 and hh is taken as |u|^2 / 2 from the u
        // actually used: then u = 1e-145 e0 and H = I - 2 e0 e0^T, a reflection - still an orthogonal similarity, and every
        // intermediate stays finite for |A| < 1e160 (p = A u / hh is 2 A[:, 0] 1e145).  e2[k] below keeps the unfloored value.
        double nrm = sqrt_nz(max_raw(nn, 1e-290));
        double u[n];
        u[0] = alpha + copysign_d(nrm, alpha);
        static_for<n - 1>([&](auto t) { u[decltype(t)::value + 1] = m[tri(k + 2 + decltype(t)::value, k)]; });
        double ihalf = rcp(__builtin_fma(u[0], u[0], sig));       // 1 / |u|^2
        double inv_hh = ihalf + ihalf;
#endif
        // p = A22 u / hh   (A22 symmetric, lower stored)
        double p[n];
        static_for<n>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            double acc = 0.0;
            static_for<n>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int hi = r > c ? r : c, lo = r > c ? c : r;
                acc = __builtin_fma(m[tri(k + 1 + hi, k + 1 + lo)], u[c], acc);
            });
            p[r] = acc * inv_hh;
        });
        double up = 0.0;
        static_for<n>([&](auto rr) { up = __builtin_fma(u[decltype(rr)::value], p[decltype(rr)::value], up); });
        double kap = 0.5 * up * inv_hh;
        // q = p - kap u ;  A22 -= u q^T + q u^T
        static_for<n>([&](auto rr) { p[decltype(rr)::value] = __builtin_fma(-kap, u[decltype(rr)::value], p[decltype(rr)::value]); });
        static_for<n>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            static_for<r + 1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                double v = m[tri(k + 1 + r, k + 1 + c)];
                v = __builtin_fma(-u[r], p[c], v);
                v = __builtin_fma(-p[r], u[c], v);
                m[tri(k + 1 + r, k + 1 + c)] = v;
            });
        });
        dg[k] = m[tri(k, k)];
        e2[k] = nn;
    });
    if constexpr (D >= 2) {
        dg[D - 2] = m[tri(D - 2, D - 2)];
        double e = m[tri(D - 1, D - 2)];
        e2[D - 2] = e * e;
    }
    dg[D - 1] = m[tri(D - 1, D - 1)];
    e2[D - 1] = 0.0;
}

#ifndef GABO_QL_RCP
// Reciprocal used inside the QL sweep: rcp_nr1 (seed + one Newton step, 2^-47) or rcp (seed + cubic correction, ~1 ulp).
// t = 1 / (p r) only scales the plane rotation of a step (c = p^2 t, s = b p t, c + s = 1 up to its error), i.e. a relative
// perturbation of 7e-15 per step on top of the recurrence's own rounding.  Measured on the benchmark distribution (N = 4096, d = 10,
// tools/ab_pairwise.py): worst relative error of d^2 against LAPACK 1.5e-13 instead of 1.0e-13, kernel 2.72 -> 2.66 ms.
#define GABO_QL_RCP rcp_nr1
#endif

// sqrt(x), x > 0, to ~2^-46: hardware seed + one coupled Goldschmidt step.  For the Wilkinson SHIFT only - its accuracy sets the
// convergence rate of a sweep, never the eigenvalues (three instructions less per sweep than sqrt_nz).
__device__ __forceinline__ double sqrt_shift(double x) {
#if defined(GABO_QL_EXACT_SHIFT_SQRT)
    return sqrt_nz(x);
#elif !defined(GABO_QL_SHIFT_SQRT_GOLDSCHMIDT)
    // the bare hardware seed (2^-24): the shift only sets the convergence RATE - a shift that misses the Wilkinson value by 1e-7 of the root
    // still contracts e^2 by ~1e-14 per sweep once the cubic phase has brought it there.  Wave-level simulation (tools/sim/ql_lookahead_sim.py
    // with the root perturbed): 124.3 -> 125.7 sweep steps and 19.3 -> 19.7 sweeps per wave for 4 instructions less per sweep; measured
    // 2.455 -> 2.440 ms.  (-DGABO_QL_SHIFT_SQRT_GOLDSCHMIDT: seed + one coupled step, 2^-46.)
    return x * __builtin_amdgcn_rsq(x);
#else
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
#endif
}

// p = gamma^2 / c feeds the next rotation as a divisor and must never be exactly 0 (a shift that hit an eigenvalue exactly gives
// gamma = 0): floor it at 1e-150, so that the product p r below stays a normal number - a perturbation far below rounding.
// Replaces LAPACK's `c == 0` branch.  One v_max_f64 per step (the first version nudged gamma itself to +-1e-75: two instructions).
__device__ __forceinline__ double floor_p(double p) {
#ifdef GABO_QL_GAMMA_NUDGE
    return p;
#else
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(p), "s"(1e-150));      // (p is a product of finite numbers: no NaN to canonicalise)
    return r;
#endif
}
// a * b + 1e-150: the same guarantee folded into the multiplication that produces p (p = a b >= 0 here) - one instruction instead of
// the product and the maximum.  The addend only registers when p < 1e-134.
__device__ __forceinline__ double mul_floor_p(double a, double b) {
#if defined(GABO_QL_GAMMA_NUDGE)
    return a * b;
#elif defined(GABO_QL_FLOOR_MAX)          /* A/B: product, then v_max_f64 */
    return floor_p(a * b);
#else
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(1e-150));
    return r;
#endif
}
__device__ __forceinline__ double nonzero(double g) {
#ifdef GABO_QL_GAMMA_NUDGE
    double a = __builtin_fmax(__builtin_fabs(g), 1e-75);
    return copysign_d(a, g);
#else
    return g;
#endif
}

// Eigenvalues of the symmetric tridiagonal (dg, sqrt(e2)) in place in dg (unordered).  Root-free QL, Wilkinson shift.
//
// Lane-uniform structure on purpose: stage l deflates e2[l]; every sweep of stage l runs the full recurrence from
// D-2 down to l with NO per-lane interior-split search, so the unrolled steps carry no predicates or branches and the
// only divergence is the iteration count per stage.  An interior off-diagonal that is (or becomes) negligible is
// simply swept through: the recurrence restarts by itself there (c -> 1, s -> 0).  p > 0 is an invariant (see
// `floor_p`), hence r = p + bb > 0 and c = p / r > 0: no division can see a zero.  The last 2x2 block is closed form.
template <int D>
__device__ __forceinline__ void tridiag_eigenvalues(double (&dg)[D], double (&e2)[D], const double eps2_arg = 0.0) {
    // Deflation threshold on e2[l] / |d[l] d[l+1]|.  LAPACK uses eps^2 (4.9e-32); 1e-20 is enough here: dropping an
    // off-diagonal e perturbs a SYMMETRIC function of the eigenvalues (sum log^2) only to second order, ~e^2 f'' <= 1e-20,
    // whatever the gap, and the iteration converges cubically, so the looser test saves the last sweep of many stages
    // (measured: 1e-32 -> 1e-22: -6 % sweep steps; 1e-22 -> 1e-20: another -1 % of the kernel's time; worst-case error of d^2 on the
    // benchmark distribution 1e-13 in all three, tools/sim/ql_lookahead_sim.py).
#ifndef GABO_QL_EPS2
#define GABO_QL_EPS2 1e-20
#endif
    // eps2_arg > 0: the caller's threshold (wave-uniform).  The Gaussian kernel VALUE without a distance output tolerates a looser one: dropping
    // an off-diagonal e^2 <= eps2 |d d'| perturbs sum log^2 lambda by ~eps2 in ABSOLUTE terms (second order), i.e. K = exp(-beta d^2) by
    // beta eps2 relative - far below its rounding at 1e-16; what the strict 1e-20 protects is the RELATIVE accuracy of a tiny distance
    // (nearly identical pairs: d ~ 1e-5, d^2 ~ 1e-10), which only the distance and Laplace outputs expose (spd_pairwise_body.hpp).
    const double eps2 = eps2_arg > 0.0 ? eps2_arg : GABO_QL_EPS2;
#ifndef GABO_QL_NOFLIP
    // QL deflates at the top (index 0) and converges fastest when the small end of a graded matrix sits there (LAPACK's
    // dsterf chooses QL vs QR on the same criterion): reverse the arrays per lane when |d[0]| > |d[D-1]|.  Measured on the
    // benchmark distribution: -9 % QL sweep steps per wave for ~40 selects.
    if constexpr (D >= 3) {
        const bool flip = __builtin_fabs(dg[0]) > __builtin_fabs(dg[D - 1]);
        static_for<D / 2>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            double a = dg[i], b = dg[D - 1 - i];
            dg[i] = flip ? b : a;
            dg[D - 1 - i] = flip ? a : b;
        });
        static_for<(D - 1) / 2>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            double a = e2[i], b = e2[D - 2 - i];
            e2[i] = flip ? b : a;
            e2[D - 2 - i] = flip ? a : b;
        });
    }
#endif
    static_for<D - 2>([&](auto ll) {
        constexpr int l = decltype(ll)::value;
        for (int it = 0; it < 60; ++it) {
#ifdef GABO_QL_NO_LOOKAHEAD
            if (e2[l] <= eps2 * __builtin_fabs(dg[l] * dg[l + 1])) break;
            const double sa = dg[l], sb = dg[l + 1], se = e2[l];
#else
            // Look-ahead shift.  The 64 lanes of a wave share the instruction stream, so stage l lasts until the SLOWEST lane has
            // deflated e2[l]; a lane that is done would idle through the other lanes' sweeps (measured on the benchmark
            // distribution: 180 sweep steps per wave against 128 per lane).  Instead it keeps sweeping with the wave - same
            // extent D-2 .. l, no per-lane masks in the unrolled steps - but takes its Wilkinson shift from block l+1, i.e. it
            // already works on its NEXT stage.  Its deflated e2[l] is set to exactly 0 first: then the steps below block l+1 pass
            // through as the identity (bb = 0 => c = 1, s = 0) whatever p has become; with a merely negligible e2[l] a tiny p at
            // the end of a converged sweep makes s = bb / (p + bb) large and the pass-through re-couples the deflated row
            // (tools/sim/ql_lookahead_sim.py: 180 -> 189 steps without the zeroing, 180 -> 129 with it).
            const bool done0 = e2[l] <= eps2 * __builtin_fabs(dg[l] * dg[l + 1]);
            // the WAVE leaves stage l when its last lane has deflated e2[l] (a per-lane `break` would keep the wave here until every lane
            // had finished its look-ahead work too, at stage l's longer sweep extent)
#ifdef GABO_QL_BALLOT_NOT
            if (__builtin_amdgcn_ballot_w64(!done0) == 0) break;
#else
            // (every active lane done: the mask of `done0` against the mask of the active lanes - one vector compare; `ballot(!done0)`
            // costs a second, NaN-aware one)
            if (__builtin_amdgcn_ballot_w64(done0) == __builtin_amdgcn_ballot_w64(true)) break;
#endif
            const double e2l = done0 ? 0.0 : e2[l];        // (e2[l] itself is rewritten at the end of the sweep: s p = 0 for these lanes)
            double sa = dg[l], sb = dg[l + 1], se = e2l;
            bool idle = false;
            if constexpr (l + 1 <= D - 3) {
                const bool done1 = e2[l + 1] <= eps2 * __builtin_fabs(dg[l + 1] * dg[l + 2]);
                idle = done0 && done1;                      // nothing to do within the look-ahead window: sit this sweep out
                sa = done0 ? dg[l + 1] : sa;
                sb = done0 ? dg[l + 2] : sb;
                se = done0 ? e2[l + 1] : se;
            } else {
                idle = done0;
            }
            if (idle) continue;
#endif
            // Wilkinson shift from the leading 2x2: sigma = d_l - e2_l / (delta + sign(delta) sqrt(delta^2 + e2_l)),
            // evaluated division-free as d_l - sign(delta) (sqrt(delta^2 + e2_l) - |delta|).  The cancellation of the
            // rationalised form only costs ~eps |delta| in the SHIFT, which changes the convergence rate, never the result.
            double delta = 0.5 * (sb - sa);
            double root = sqrt_shift(__builtin_fma(delta, delta, se));
            double sigma = sa - copysign_d(root - __builtin_fabs(delta), delta);
            double gamma = nonzero(dg[D - 1] - sigma);
            double p = mul_floor_p(gamma, gamma);
            double s = 0.0;
#ifdef GABO_QL_FORM1
            double c = 1.0;
#endif
            static_for_down<D - 2, l>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
#ifdef GABO_QL_NO_LOOKAHEAD
                double bb = e2[i];
#else
                double bb = (i == l) ? e2l : e2[i];
#endif
                double r = p + bb;
                if constexpr (i != D - 2) e2[i + 1] = s * r;
#ifdef GABO_QL_FORM1
                // (round-1 form: c and s explicitly, gamma = c (a_i - sigma) - s gamma_old, p' = (gamma r)^2 t)
                double t = GABO_QL_RCP(p * r);
                double ir = t * p;
                c = p * ir;
                s = bb * ir;
                double oldgam = gamma;
                double al = dg[i];
                gamma = nonzero(__builtin_fma(c, al - sigma, -s * oldgam));
                dg[i + 1] = oldgam + (al - gamma);
                double gr = gamma * r;
                p = floor_p((gr * t) * gr);
#else
                // one reciprocal serves the step: t = 1/(p r)  =>  1/r = t p.  With f = p (a_i - sigma) - b gamma_old:
                //   gamma' = c (a_i - sigma) - s gamma_old = f / r = f (t p),     p' = gamma'^2 / c = gamma'^2 r / p = f^2 t
                // (c itself is never needed; two instructions less per step than forming c, gamma r and (gamma r)^2 t)
                double t = GABO_QL_RCP(p * r);
                double ir = t * p;
                s = bb * ir;
                double oldgam = gamma;
                double al = dg[i];
                double f = __builtin_fma(p, al - sigma, -(bb * oldgam));
                gamma = ir * f;
                dg[i + 1] = oldgam + (al - gamma);
                p = mul_floor_p(f * t, f);
#endif
            });
            e2[l] = s * p;
            // (a look-ahead lane's deflated d_l comes back as sigma + (d_l - sigma): a perturbation of an ulp of |sigma| <= |T|, the
            // size of the backward error of every other step of the sweep)
            dg[l] = sigma + gamma;
        }
    });
    // trailing 2x2 [[a, b], [b, c]]: rt1 = larger-magnitude root, rt2 = det / rt1
    {
        double a = dg[D - 2], b2 = e2[D - 2], cc = dg[D - 1];
        double sm = a + cc, df = a - cc;
        double rt = sqrt_pos(__builtin_fma(df, df, 4.0 * b2));
        double r1 = 0.5 * (sm + copysign_d(rt, sm));
        double det = __builtin_fma(a, cc, -b2);
        dg[D - 2] = r1;
        dg[D - 1] = (r1 == 0.0) ? 0.0 : det * rcp(r1);
    }
}

}  // namespace gabo
