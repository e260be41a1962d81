// The nested projection Y = W^T X W (nested_mappings/nested_spd_utils.py:13-48) as ONE linear map on Mandel vectors: y = P x with the
// dl_vec x D_vec matrix P built from W (shared by gabo_spd_project and the fused preparation of the nested Gram matrices).
#pragma once
#include "gabo_device.hpp"

namespace gabo {

// Wl: D x DL (LDS copy of w), P: DV x Dv (LDS), DV = DL (DL + 1) / 2, Dv = D (D + 1) / 2.  Called by every thread of the block; ends with a barrier.
template <int DL>
__device__ __forceinline__ void build_projection_operator(const double* __restrict__ w, int D, double* Wl, double* P) {
    const int Dv = D * (D + 1) / 2;
    for (int e = threadIdx.x; e < D * DL; e += blockDim.x) Wl[e] = w[e];
    __syncthreads();
    for (int t = threadIdx.x; t < D * D; t += blockDim.x) {
        const int r = t / D, c = t - r * D;
        if (r < c) continue;
        const int k = r - c, idx = mandel_pos(D, r, c);      // input entry (r, c) sits on sub-diagonal k of the Mandel order
        static_for<DL>([&](auto kk) {
            constexpr int ko = decltype(kk)::value;
            static_for<DL - ko>([&](auto bb) {
                constexpr int b = decltype(bb)::value, a = b + ko;
                double coef;
                if (k == 0) coef = Wl[r * DL + a] * Wl[r * DL + b] * (ko == 0 ? 1.0 : kSqrt2);
                else coef = __builtin_fma(Wl[r * DL + a], Wl[c * DL + b], Wl[c * DL + a] * Wl[r * DL + b]) * (ko == 0 ? kInvSqrt2 : 1.0);
                P[mandel_pos(DL, a, b) * Dv + idx] = coef;
            });
        });
    }
    __syncthreads();
}

// one matrix per wave: y = P x, every lane ends with all DV sums (the same bits in every lane)
template <int DL>
__device__ __forceinline__ void project_one(const double* __restrict__ xi, const double* P, int Dv, double (&y)[DL * (DL + 1) / 2]) {
    constexpr int DV = DL * (DL + 1) / 2;
    const int lane = threadIdx.x & 63;
    double acc[DV];
    static_for<DV>([&](auto e) { acc[decltype(e)::value] = 0.0; });
    for (int k = lane; k < Dv; k += 64) {
        const double xk = xi[k];
        static_for<DV>([&](auto e) { acc[decltype(e)::value] = __builtin_fma(P[decltype(e)::value * Dv + k], xk, acc[decltype(e)::value]); });
    }
    static_for<DV>([&](auto e) {
        double s = acc[decltype(e)::value];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        y[decltype(e)::value] = s;
    });
}

}  // namespace gabo
