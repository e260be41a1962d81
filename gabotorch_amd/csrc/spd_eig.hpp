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
        double nrm = sqrt_pos(nn);
        double u[n];
        u[0] = alpha + copysign_d(nrm, alpha);
        static_for<n - 1>([&](auto t) { u[decltype(t)::value + 1] = m[tri(k + 2 + decltype(t)::value, k)]; });
        double hh = __builtin_fma(__builtin_fabs(alpha), nrm, nn);
        double inv_hh = hh == 0.0 ? 0.0 : rcp(hh);
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

// Eigenvalues of the symmetric tridiagonal (dg, sqrt(e2)) in place in dg (unordered).  Root-free QL with
// Wilkinson-type shift; per-lane deflation index m is data dependent and handled by predication.
template <int D>
__device__ __forceinline__ void tridiag_eigenvalues(double (&dg)[D], double (&e2)[D]) {
    constexpr double eps = 2.220446049250313e-16;
    constexpr double eps2 = eps * eps;
    static_for<D - 1>([&](auto ll) {
        constexpr int l = decltype(ll)::value;
        for (int it = 0; it < 40; ++it) {
            // m = first index >= l whose off-diagonal is negligible (D-1 if none)
            int mi = D - 1;
            static_for_down<D - 2, l>([&](auto mm) {
                constexpr int q = decltype(mm)::value;
                if (e2[q] <= eps2 * __builtin_fabs(dg[q] * dg[q + 1])) mi = q;
            });
            if (mi == l) break;
            // shift from the leading 2x2 of the unreduced block
            double pl = dg[l];
            double rte = sqrt_pos(e2[l]);
            double sg = (dg[l + 1] - pl) * rcp(2.0 * rte);
            double rr = sqrt_pos(__builtin_fma(sg, sg, 1.0));
            double sigma = pl - rte * rcp(sg + copysign_d(rr, sg));
            double c = 1.0, s = 0.0, gamma = 0.0, p = 0.0;
            static_for_down<D - 2, l>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                if (i < mi) {
                    if (i == mi - 1) {
                        gamma = dg[i + 1] - sigma;
                        p = gamma * gamma;
                    }
                    double bb = e2[i];
                    double r = p + bb;
                    if (i != mi - 1) e2[i + 1] = s * r;
                    double oldc = c;
                    double ir = rcp(r);
                    c = p * ir;
                    s = bb * ir;
                    double oldgam = gamma;
                    double al = dg[i];
                    gamma = __builtin_fma(c, al - sigma, -s * oldgam);
                    dg[i + 1] = oldgam + (al - gamma);
                    p = (c != 0.0) ? gamma * gamma * rcp(c) : oldc * bb;
                }
            });
            e2[l] = s * p;
            dg[l] = sigma + gamma;
        }
    });
}

}  // namespace gabo
