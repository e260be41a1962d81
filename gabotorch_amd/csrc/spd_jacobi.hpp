// Per-lane cyclic Jacobi eigen-decomposition (eigenvalues AND eigenvectors) of a small symmetric matrix held in registers:
// shared by the SPD backward kernel (spd_backward.hip) and the fused acquisition kernel (spd_acq.hip).
#pragma once
#include "gabo_device.hpp"

namespace gabo {

// Cyclic Jacobi on a packed lower triangle (registers) with accumulated eigenvectors kept in LDS: the D*D doubles of V
// per lane do not fit the 256 directly addressable VGPRs next to M.  vl[(r*D + c)*64 + lane], columns = eigenvectors.
// V(r, c) -> double& : where the eigenvector entries live (LDS column of the lane, or a register array); every call site passes
// compile-time r, c, so a register-array accessor resolves to fixed registers after inlining.
template <int D, class VAcc>
__device__ __forceinline__ void jacobi_eig_acc(double (&m)[tri_size(D)], VAcc&& V) {
    static_for<D>([&](auto rr) {
        static_for<D>([&](auto cc) {
            V(decltype(rr)::value, decltype(cc)::value) = (decltype(rr)::value == decltype(cc)::value) ? 1.0 : 0.0;
        });
    });
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, dia = 0.0;
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            dia = __builtin_fma(m[tri(r, r)], m[tri(r, r)], dia);
            static_for<r>([&](auto cc) { double x = m[tri(r, decltype(cc)::value)]; off = __builtin_fma(x, x, off); });
        });
        if (off <= 1e-33 * dia) break;
        static_for<D - 1>([&](auto pp) {
            constexpr int p = decltype(pp)::value;
            static_for<D - 1 - p>([&](auto qq) {
                constexpr int q = p + 1 + decltype(qq)::value;
                double apq = m[tri(q, p)];
                double app = m[tri(p, p)], aqq = m[tri(q, q)];
                // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)),  theta = (aqq - app) / (2 apq); written division-safe:
                // t = 2 |apq| sgn(apq h) / (|h| + sqrt(h^2 + 4 apq^2)),  h = aqq - app
                double h = aqq - app;
                double den = __builtin_fabs(h) + sqrt_pos(__builtin_fma(h, h, 4.0 * apq * apq));
                double t = (den == 0.0) ? 0.0 : copysign_d(2.0 * apq, apq * h) * rcp(den == 0.0 ? 1.0 : den);
                if (h == 0.0) t = (apq == 0.0) ? 0.0 : copysign_d(1.0, apq);
                double c = rsqrt_nz(__builtin_fma(t, t, 1.0));
                double s = t * c;
                m[tri(p, p)] = __builtin_fma(-t, apq, app);
                m[tri(q, q)] = __builtin_fma(t, apq, aqq);
                m[tri(q, p)] = 0.0;
                static_for<D>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    if constexpr (k != p && k != q) {
                        constexpr int ikp = k > p ? tri(k, p) : tri(p, k);
                        constexpr int ikq = k > q ? tri(k, q) : tri(q, k);
                        double akp = m[ikp], akq = m[ikq];
                        m[ikp] = __builtin_fma(c, akp, -s * akq);
                        m[ikq] = __builtin_fma(s, akp, c * akq);
                    }
                    double vkp = V(k, p), vkq = V(k, q);
                    V(k, p) = __builtin_fma(c, vkp, -s * vkq);
                    V(k, q) = __builtin_fma(s, vkp, c * vkq);
                });
            });
        });
    }
}

// eigenvectors in LDS: vl[(r*D + c)*64] (one column of 64-lane-strided entries per lane)
template <int D>
__device__ __forceinline__ void jacobi_eig(double (&m)[tri_size(D)], double* __restrict__ vl) {
    jacobi_eig_acc<D>(m, [&](int r, int c) -> double& { return vl[(r * D + c) * 64]; });
}

// eigenvectors in registers (latency-bound callers with few waves per SIMD and D small enough: D*D + D(D+1)/2 doubles live)
template <int D>
__device__ __forceinline__ void jacobi_eig_reg(double (&m)[tri_size(D)], double (&v)[D * D]) {
    jacobi_eig_acc<D>(m, [&](int r, int c) -> double& { return v[r * D + c]; });
}

}  // namespace gabo
