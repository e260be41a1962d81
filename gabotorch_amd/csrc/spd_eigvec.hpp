// Per-lane symmetric eigen-DECOMPOSITION (eigenvalues and eigenvectors) of a small fixed-size matrix, everything in the
// lane's registers: Householder tridiagonalisation, explicit accumulation of the reflectors, implicit-shift QL with the plane
// rotations applied to the eigenvector columns (the EISPACK tred2 + tql2 / LAPACK dsytrd + dorgtr + dsteqr scheme).
//
// Used wherever a matrix FUNCTION is needed per lane: the SPD backward kernel (logm(M) = Z diag(log lambda) Z^T per pair), the fused
// acquisition evaluation and the trust-region kernels (logm, expm of tangent vectors, extreme eigenpairs).  The cyclic Jacobi it
// replaced (spd_jacobi.hpp, still used with V in LDS by the acquisition kernels above d = 8) costs ~4e4 instructions per matrix at
// d = 10 and is one long dependent chain; this one is ~1.2e4 with Z in registers (200 VGPRs at d = 10: one wave per SIMD, 512
// VGPRs) and no LDS traffic in the iteration.  One lane owns one matrix; all register-array indices are compile-time
// constants.  Like the eigenvalue-only solver (spd_eig.hpp) every sweep of stage l runs the whole recurrence D-2 .. l without an
// interior split search, and a lane that has deflated its stage pre-converges the next one while its wave finishes (look-ahead).
#pragma once
#include "gabo_device.hpp"

namespace gabo {

// m: packed lower triangle (destroyed).  Out: dg (diagonal), e (signed sub-diagonal, e[D-1] = 0) of T = Q^T M Q and z = Q
// (row-major D x D), Q = H_0 H_1 ... H_{D-3}.
// First half: T = Q^T M Q with the reflectors left in the columns of m (u_k in m[k+1.., k]) and 1 / hh_k in ihh.
template <int D>
__device__ __forceinline__ void tridiagonalize_reflectors(double (&m)[tri_size(D)], double (&dg)[D], double (&e)[D],
                                                          double (&ihh)[D >= 3 ? D - 2 : 1]) {
    static_for<D - 2>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int n = D - k - 1;  // order of the trailing block; the column below the diagonal is x_0..x_{n-1}
        // u = x + sign(x0)|x| e0 ;  H = I - u u^T / hh,  hh = |x|^2 + |x0||x| ;  H x = -sign(x0)|x| e0
        double alpha = m[tri(k + 1, k)];
        double sig = 0.0;
        static_for<n - 1>([&](auto t) { double x = m[tri(k + 2 + decltype(t)::value, k)]; sig = __builtin_fma(x, x, sig); });
        double nn = __builtin_fma(alpha, alpha, sig);
        double nrm = sqrt_pos(nn);
        double u[n];
        u[0] = alpha + copysign_d(nrm, alpha);
        static_for<n - 1>([&](auto t) { u[decltype(t)::value + 1] = m[tri(k + 2 + decltype(t)::value, k)]; });
        double hh = __builtin_fma(__builtin_fabs(alpha), nrm, nn);
        double inv_hh = hh == 0.0 ? 0.0 : rcp(hh);
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
        e[k] = hh == 0.0 ? alpha : -copysign_d(nrm, alpha);   // (hh == 0: the column is already zero, H = I)
        m[tri(k + 1, k)] = u[0];                                 // the reflector stays in column k for the accumulation below
        ihh[k] = inv_hh;
    });
    if constexpr (D >= 2) {
        dg[D - 2] = m[tri(D - 2, D - 2)];
        e[D - 2] = m[tri(D - 1, D - 2)];
    }
    dg[D - 1] = m[tri(D - 1, D - 1)];
    e[D - 1] = 0.0;
}

template <int D>
__device__ __forceinline__ void tridiagonalize_q(double (&m)[tri_size(D)], double (&dg)[D], double (&e)[D], double (&z)[D * D]) {
    double ihh[D >= 3 ? D - 2 : 1];
    tridiagonalize_reflectors<D>(m, dg, e, ihh);
    // z = H_0 ... H_{D-3}: identity, then the reflectors from the last to the first; H_k touches rows k+1.. only, and what has been
    // built so far is non-trivial in rows / columns >= k+2, so each step fills the block [k+1.., k+1..] (everything else stays 0 / 1
    // and folds away at compile time)
    static_for<D>([&](auto rr) {
        static_for<D>([&](auto cc) { z[decltype(rr)::value * D + decltype(cc)::value] = (decltype(rr)::value == decltype(cc)::value) ? 1.0 : 0.0; });
    });
    static_for_down<D - 3, 0>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int n = D - k - 1;
        static_for<n>([&](auto cc) {
            constexpr int c = k + 1 + decltype(cc)::value;
            double t = 0.0;
            static_for<n>([&](auto rr) {
                constexpr int r = k + 1 + decltype(rr)::value;
                t = __builtin_fma(m[tri(r, k)], z[r * D + c], t);
            });
            t *= ihh[k];
            static_for<n>([&](auto rr) {
                constexpr int r = k + 1 + decltype(rr)::value;
                z[r * D + c] = __builtin_fma(-m[tri(r, k)], t, z[r * D + c]);
            });
        });
    });
}

// Eigen-decomposition of the symmetric tridiagonal (dg, e) with the rotations accumulated into z (columns become the eigenvectors of
// the ORIGINAL matrix when z enters as the Q of tridiagonalize_q).  Eigenvalues in dg, unordered.  ROWS: the rows of z this lane holds (all D
// of them, or its share when several lanes split one matrix by rows: a rotation acts on two COLUMNS, so it never mixes rows).
// VEC = false: the same recurrence on (dg, e) without touching z - the eigenvalues come out with the SAME BITS as with the vectors (the
// rotations never feed back into dg / e), at about a third of the instructions.
template <int D, int ROWS = D, bool VEC = true>
__device__ __forceinline__ void tridiag_ql_vectors(double (&dg)[D], double (&e)[D], double (&z)[ROWS * D], const double eps2_arg = 0.0) {
    // dropping an off-diagonal e perturbs a matrix FUNCTION to first order in e / gap (the eigenvalue-only solver can be looser: a
    // symmetric function of the eigenvalues moves only to second order): |e| <= eps (|d_l| + |d_{l+1}|), the EISPACK / LAPACK
    // criterion in the form that also terminates on indefinite input (tangent vectors) with zeros on the diagonal
    // (eps2_arg > 0: the caller's threshold on e^2 / (|d_l| + |d_l+1|)^2.  A dropped off-diagonal e changes a matrix function by e f[lambda_l, lambda_l+1] - a divided
    // difference, bounded by e f' even where the gap closes - so a caller that needs the function to 1e-12 can stop at |e| <= 1e-13 (|d| + |d'|))
    const double eps2 = eps2_arg > 0.0 ? eps2_arg : 1.2e-32;
    static_for<D - 1>([&](auto ll) {
        constexpr int l = decltype(ll)::value;
        for (int it = 0; it < 60; ++it) {
            const double s01 = __builtin_fabs(dg[l]) + __builtin_fabs(dg[l + 1]);
            const bool done0 = e[l] * e[l] <= eps2 * (s01 * s01);
            if (__builtin_amdgcn_ballot_w64(done0) == __builtin_amdgcn_ballot_w64(true)) break;   // the wave leaves the stage with its last lane (one compare: see spd_eig.hpp)
            const double el = done0 ? 0.0 : e[l];
            double sa = dg[l], sb = dg[l + 1], se = el;
            bool idle = done0;
            if constexpr (l + 1 <= D - 2) {
                // look-ahead (see spd_eig.hpp): a lane that is done with stage l takes its shift from block l+1; its e[l] is exactly 0,
                // so the steps below that block are identities (c = +-1, s = 0)
                const double s12 = __builtin_fabs(dg[l + 1]) + __builtin_fabs(dg[l + 2]);
                const bool done1 = e[l + 1] * e[l + 1] <= eps2 * (s12 * s12);
                idle = done0 && done1;
                sa = done0 ? dg[l + 1] : sa;
                sb = done0 ? dg[l + 2] : sb;
                se = done0 ? e[l + 1] : se;
            }
            if (idle) continue;
            // Wilkinson shift from the leading 2x2 [[sa, se], [se, sb]]: sigma = sa - se / (theta + sign(theta) sqrt(theta^2 + 1)),
            // theta = (sb - sa) / (2 se); written without the division by se: sigma = sa - se^2 / (delta + sign(delta) sqrt(delta^2 + se^2))
            double delta = 0.5 * (sb - sa);
            double root = sqrt_nz(__builtin_fma(delta, delta, se * se));
            double sigma = sa - copysign_d(root - __builtin_fabs(delta), delta);
            double g = dg[D - 1] - sigma;
            double s = 1.0, c = 1.0, p = 0.0;
            static_for_down<D - 2, l>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const double ei = (i == l) ? el : e[i];
                double f = s * ei;
                double b = c * ei;
                // g must not vanish together with f (r = 0): nudge an exact zero to +-1e-150, far below rounding
                g = copysign_d(__builtin_fmax(__builtin_fabs(g), 1e-150), g);
                double r2 = __builtin_fma(f, f, g * g);
                double ir = rsqrt_nz(r2);
                e[i + 1] = r2 * ir;
                s = f * ir;
                c = g * ir;
                g = dg[i + 1] - p;
                double r = __builtin_fma(dg[i] - g, s, 2.0 * c * b);
                p = s * r;
                dg[i + 1] = g + p;
                g = __builtin_fma(c, r, -b);
                if constexpr (VEC) {
                    static_for<ROWS>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
                        double zi = z[k * D + i], zj = z[k * D + i + 1];
                        z[k * D + i + 1] = __builtin_fma(s, zi, c * zj);
                        z[k * D + i] = __builtin_fma(c, zi, -s * zj);
                    });
                }
            });
            dg[l] -= p;
            e[l] = g;
            e[D - 1] = 0.0;
        }
    });
}

// The vector pass of the TWO-PASS decomposition (sym_eig_reg_two_pass below): the same implicit QL step, with the rotations accumulated into z, but
// the first sweep of stage l shifted by `known[l]` - the eigenvalue an eigenvalue-only pass over the same tridiagonal matrix left at position l.
// Shifted by an exact eigenvalue, ONE sweep deflates e[l] (in exact arithmetic to zero; in fp64, on the benchmark distribution, to below the
// threshold in 99.7 % of the lane-stages: tools/sim/perfect_shift_sim.py); a stage whose test fails in some lane goes on with Wilkinson shifts as
// the one-pass solver does, so the result never depends on the shortcut having worked.  No look-ahead: every lane starts stage l at its first sweep.
template <int D, int ROWS = D>
__device__ __forceinline__ void tridiag_ql_vectors_known(double (&dg)[D], double (&e)[D], double (&z)[ROWS * D], const double (&known)[D],
                                                         const double eps2_arg = 0.0) {
    const double eps2 = eps2_arg > 0.0 ? eps2_arg : 1.2e-32;
    static_for<D - 1>([&](auto ll) {
        constexpr int l = decltype(ll)::value;
        for (int it = 0; it < 60; ++it) {
            const double s01 = __builtin_fabs(dg[l]) + __builtin_fabs(dg[l + 1]);
            const bool done0 = e[l] * e[l] <= eps2 * (s01 * s01);
            if (__builtin_amdgcn_ballot_w64(done0) == __builtin_amdgcn_ballot_w64(true)) break;
            if (done0) continue;
            double sigma;
            if (it == 0) {
                sigma = known[l];
            } else {
                const double sa = dg[l], sb = dg[l + 1], se = e[l];
                const double delta = 0.5 * (sb - sa);
                const double root = sqrt_nz(__builtin_fma(delta, delta, se * se));
                sigma = sa - copysign_d(root - __builtin_fabs(delta), delta);
            }
            double g = dg[D - 1] - sigma;
            double s = 1.0, c = 1.0, p = 0.0;
            static_for_down<D - 2, l>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const double ei = e[i];
                double f = s * ei;
                double b = c * ei;
                g = copysign_d(__builtin_fmax(__builtin_fabs(g), 1e-150), g);
                double r2 = __builtin_fma(f, f, g * g);
                double ir = rsqrt_nz(r2);
                e[i + 1] = r2 * ir;
                s = f * ir;
                c = g * ir;
                g = dg[i + 1] - p;
                double r = __builtin_fma(dg[i] - g, s, 2.0 * c * b);
                p = s * r;
                dg[i + 1] = g + p;
                g = __builtin_fma(c, r, -b);
                static_for<ROWS>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    double zi = z[k * D + i], zj = z[k * D + i + 1];
                    z[k * D + i + 1] = __builtin_fma(s, zi, c * zj);
                    z[k * D + i] = __builtin_fma(c, zi, -s * zj);
                });
            });
            dg[l] -= p;
            e[l] = g;
            e[D - 1] = 0.0;
        }
    });
}

// m = V diag(lam) V^T like sym_eig_reg, in TWO passes over the tridiagonal matrix: the eigenvalues first (the O(1)-per-rotation recurrence alone,
// with its look-ahead), then the rotations that build V with every stage shifted by its known eigenvalue - one sweep per stage, D (D - 1) / 2
// rotations on the vectors instead of the ~2 sweeps per eigenvalue of the slowest lane the one-pass solver applies to them.  A rotation with
// vectors costs 17 + 4 D instructions, without 17: at D = 10 and 130 sweep steps per wave the QL part goes from ~7.4 k to ~2.2 k + ~2.9 k
// instructions per pair.  lam: the eigenvalues the vector pass ends with (they go with V; they differ from the first pass's by rounding).
template <int D>
__device__ __forceinline__ void sym_eig_reg_two_pass(double (&m)[tri_size(D)], double (&lam)[D], double (&v)[D * D], const double eps2 = 0.0) {
    double sub[D], known[D], e1[D], none[D];
    tridiagonalize_q<D>(m, lam, sub, v);
    static_for<D>([&](auto kk) {
        known[decltype(kk)::value] = lam[decltype(kk)::value];
        e1[decltype(kk)::value] = sub[decltype(kk)::value];
    });
    tridiag_ql_vectors<D, 1, false>(known, e1, none, eps2);
    tridiag_ql_vectors_known<D>(lam, sub, v, known, eps2);
}

// m (packed lower triangle, destroyed) = V diag(lam) V^T: eigenvalues (unordered) and eigenvectors (columns of v, row-major D x D).
// Every lane of the wave must call it (the iteration leaves a stage by a wave-wide vote).
template <int D>
__device__ __forceinline__ void sym_eig_reg(double (&m)[tri_size(D)], double (&lam)[D], double (&v)[D * D], const double eps2 = 0.0) {
    double sub[D];
    tridiagonalize_q<D>(m, lam, sub, v);
    tridiag_ql_vectors<D>(lam, sub, v, eps2);
}

// The eigenvalues alone, bit-identical to the `lam` of sym_eig_reg on the same input (same tridiagonalisation, same QL recurrence; the
// reflectors are not accumulated and the rotations not applied): for callers that need the VALUE of a spectral function now and its
// gradient only sometimes (the trust-region solve evaluates a proposal's acquisition value first, spd_tr_body.hpp).
template <int D>
__device__ __forceinline__ void sym_eig_reg_values(double (&m)[tri_size(D)], double (&lam)[D]) {
    double sub[D], ihh[D >= 3 ? D - 2 : 1], none[D];
    tridiagonalize_reflectors<D>(m, lam, sub, ihh);
    tridiag_ql_vectors<D, 1, false>(lam, sub, none);
}

// ---- two lanes per matrix ("duo") ----------------------------------------------------------------------------------------------------
// Lanes 2p and 2p + 1 hold the same matrix; each keeps HALF the rows of Z (lane parity h owns rows 2 lr + h, lr = 0 .. HR - 1, HR = ceil(D / 2);
// for odd D the last row of the odd lane is padding and stays zero).  Z is what makes the one-lane solver a 512-register kernel (D^2 doubles:
// 200 VGPRs at d = 10, one wave per SIMD, every instruction at 8.5 instead of 4.5 cycles); halved it fits 256 registers up to d = 12 (two waves
// per SIMD) and 512 up to d = 16.  The QL rotations - two thirds of the work - act on two columns of Z row by row, so they need no exchange at all:
// a lane simply applies them to its rows.  The tridiagonalisation and the scalar QL recurrence run redundantly (bit-identically) in both lanes; the
// accumulation of the reflectors needs one partner sum per (reflector, column) through a DPP swap.
template <int D>
struct DuoShape {
    static constexpr int HR = (D + 1) / 2;
};

// the partner lane's value (lane ^ 1)
__device__ __forceinline__ double duo_swap(double v) { return dpp_fetch<0xB1, 0xf>(v); }

// zh[lr * D + c] = Z[2 lr + h][c]
template <int D>
__device__ __forceinline__ void tridiagonalize_q_duo(double (&m)[tri_size(D)], double (&dg)[D], double (&e)[D],
                                                     double (&zh)[DuoShape<D>::HR * D], bool h) {
    constexpr int HR = DuoShape<D>::HR;
    double ihh[D >= 3 ? D - 2 : 1];
    tridiagonalize_reflectors<D>(m, dg, e, ihh);
    const double one_odd = h ? 1.0 : 0.0, one_even = h ? 0.0 : 1.0;
    static_for<HR>([&](auto rr) {
        constexpr int lr = decltype(rr)::value;
        static_for<D>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            zh[lr * D + c] = (c == 2 * lr) ? one_even : ((c == 2 * lr + 1) ? one_odd : 0.0);
        });
    });
    static_for_down<D - 3, 0>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int lr0 = k / 2;                      // local rows below it belong to global rows <= k for either parity
        // this lane's entries of the reflector (zero above row k + 1: compile-time zeros fold the products away)
        double um[HR];
        static_for<HR - lr0>([&](auto rr) {
            constexpr int lr = lr0 + decltype(rr)::value;
            constexpr int re = 2 * lr, ro = 2 * lr + 1;
            const double ue = (re >= k + 1 && re < D) ? m[tri(re < D ? re : D - 1, k)] : 0.0;
            const double uo = (ro >= k + 1 && ro < D) ? m[tri(ro < D ? ro : D - 1, k)] : 0.0;
            um[lr] = h ? uo : ue;
        });
        static_for<D - k - 1>([&](auto cc) {
            constexpr int c = k + 1 + decltype(cc)::value;
            double t = 0.0;
            static_for<HR - lr0>([&](auto rr) {
                constexpr int lr = lr0 + decltype(rr)::value;
                t = __builtin_fma(um[lr], zh[lr * D + c], t);
            });
            t = (t + duo_swap(t)) * ihh[k];
            static_for<HR - lr0>([&](auto rr) {
                constexpr int lr = lr0 + decltype(rr)::value;
                zh[lr * D + c] = __builtin_fma(-um[lr], t, zh[lr * D + c]);
            });
        });
    });
}

// m (packed lower triangle, the same in both lanes of a pair, destroyed) = V diag(lam) V^T: eigenvalues (unordered, in both lanes) and this
// lane's rows of V.  Every lane of the wave must call it.
template <int D>
__device__ __forceinline__ void sym_eig_duo(double (&m)[tri_size(D)], double (&lam)[D], double (&vh)[DuoShape<D>::HR * D], bool h) {
    double sub[D];
    tridiagonalize_q_duo<D>(m, lam, sub, vh, h);
    tridiag_ql_vectors<D, DuoShape<D>::HR>(lam, sub, vh);
}

}  // namespace gabo
