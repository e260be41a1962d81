// Symmetric eigen-decomposition of ONE d x d matrix (5 <= d <= 32) by ONE wave: the tred2 + tql2 scheme of spd_eigvec.hpp with the
// matrix spread over the lanes instead of held by one lane.  Lane r owns row r of the matrix during the Householder reduction and row r
// of the eigenvector matrix afterwards, both in registers with compile-time indices (the order is padded to DP in {8, 12, 16, ..., 32} with
// an identity block in FRONT of the matrix: the padded reflectors are skipped by a uniform branch and the padded QL stages deflate before
// their first sweep, so the padding costs no iteration).  What is wave-uniform - the reflector u_k, the tridiagonal (dg, e), the QL
// recurrence - is either broadcast through LDS (one ds_read_b64/b128 per entry, conflict-free by construction) or computed redundantly
// in every lane; a plane rotation acts on two COLUMNS of Z, so the lanes apply it to their rows without any exchange.
//
// Why not the parallel-ordering Jacobi of lds_linalg.hpp above d ~ 8: one Jacobi round is three barrier phases whose arithmetic is a
// dependent chain (angle: rsq + rcp + ~20 fp64 instructions), ~130 rounds per 20 x 20 matrix = 0.21 ms measured.
// Here the chain is the QL recurrence alone (13 instructions per rotation, ~1.7 (d^2 / 2) rotations) and the O(d^3) work is d-way
// lane-parallel: ~8e3 wave instructions at d = 20.  A lone wave issues one fp64 instruction per 8.5 cycles (profiles/r02_ubench_issue.txt).
#pragma once
#include "gabo_device.hpp"
#include "spd_eigvec.hpp"

// Development instrumentation (tools/ubench_eigh.hip builds with -DGABO_EIGH_CLOCKS): block 0 / lane 0 stores the shader clock at the
// phase boundaries.  Compiles to nothing otherwise.
#ifdef GABO_EIGH_CLOCKS
static __device__ long long gabo_eigh_clk[8];
#define GABO_EIGH_TICK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) gabo_eigh_clk[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GABO_EIGH_TICK(i) do { } while (0)
#endif

namespace gabo {

// LDS scratch (doubles) next to the matrices: the (u_j, q_j) pairs of the Householder step in flight
constexpr int kWaveEighScratch = 64;
constexpr int kWaveEighMinDim = 5;

__device__ __forceinline__ double lane_value(double v, int lane_const) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane_const),
                            __builtin_amdgcn_readlane(__double2loint(v), lane_const));
}

// LDS pointer in its own address space: a noinline function would otherwise reach its arguments through flat_load / flat_store
using lds_f64 = __attribute__((address_space(3))) double;

// LDS written by some lanes of this wave is read by others: the hardware queue is in order, this keeps the compiler in order too
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A (d x d, symmetric, row-major, LDS): on return its DIAGONAL holds the eigenvalues (unordered; the rest of A is left as it was);
// V (d x d, LDS, may be null: eigenvalues only): eigenvectors in columns; bc: kWaveEighScratch doubles of LDS.  DP - 3 <= d <= DP.
// Called by all 64 lanes of one wave (any other waves of the block wait at the caller's barrier).
template <int DP>
__device__ __attribute__((noinline)) void wave_eigh(lds_f64* A, lds_f64* V, lds_f64* bc, int d) {
    const int lane = threadIdx.x & 63;
    const int pad = DP - d;
    const int ra = lane - pad;                           // the row of A this lane owns (lanes < pad: identity rows; lanes >= DP: idle)
    const bool own = ra >= 0 && lane < DP;
    double a[DP];
    static_for<DP>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double v = (lane == c) ? 1.0 : 0.0;
        if (own && c >= pad) v = A[ra * d + (c - pad)];
        a[c] = v;
    });
    GABO_EIGH_TICK(0);
    double dg[DP], e[DP], ihh[DP >= 3 ? DP - 2 : 1];
    lds_f64* refl = V;                                    // u_k (k >= pad), entry j at refl[(k - pad) d + (j - pad)]; overwritten by Z at the end
    static_for<DP - 2>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        if (k < pad) {                                   // wave-uniform: the identity block needs no reflector
            dg[k] = 1.0;
            e[k] = 0.0;
            ihh[k] = 0.0;
            return;
        }
        const bool below = lane > k && lane < DP;
        const double x = below ? a[k] : 0.0;             // column k below the diagonal = entry k of the rows below (symmetry)
        const double alpha = lane_value(a[k], k + 1);
        const double nn = wave_allsum(x * x);
        const double nrm = sqrt_pos(nn);
        const double hh = __builtin_fma(__builtin_fabs(alpha), nrm, nn);      // u = x + sign(x0)|x| e0, H = I - u u^T / hh
        const double inv_hh = hh == 0.0 ? 0.0 : rcp(hh);
        const double u = (lane == k + 1) ? alpha + copysign_d(nrm, alpha) : x;
        if (lane < DP) bc[2 * lane] = u;
        if (V != nullptr && below) refl[(k - pad) * d + (lane - pad)] = u;
        wave_lds_order();
        double p = 0.0;
        static_for<DP - k - 1>([&](auto jj) {
            constexpr int j = k + 1 + decltype(jj)::value;
            p = __builtin_fma(a[j], bc[2 * j], p);
        });
        p = below ? p * inv_hh : 0.0;
        const double kap = 0.5 * wave_allsum(u * p) * inv_hh;
        const double q = __builtin_fma(-kap, u, p);
        if (lane < DP) bc[2 * lane + 1] = q;
        wave_lds_order();
        static_for<DP - k - 1>([&](auto jj) {
            constexpr int j = k + 1 + decltype(jj)::value;
            const double ub = bc[2 * j], qb = bc[2 * j + 1];
            a[j] = __builtin_fma(-q, ub, __builtin_fma(-u, qb, a[j]));
        });
        dg[k] = lane_value(a[k], k);
        e[k] = hh == 0.0 ? alpha : -copysign_d(nrm, alpha);
        ihh[k] = inv_hh;
        wave_lds_order();                                // the next step rewrites bc
    });
    dg[DP - 2] = lane_value(a[DP - 2], DP - 2);
    e[DP - 2] = lane_value(a[DP - 2], DP - 1);
    dg[DP - 1] = lane_value(a[DP - 1], DP - 1);
    e[DP - 1] = 0.0;
    GABO_EIGH_TICK(1);
    // row `lane` of Q = H_pad ... H_{DP-3}: e_lane^T pushed through the reflectors in order
    double z[DP];
    static_for<DP>([&](auto cc) { z[decltype(cc)::value] = (lane == decltype(cc)::value) ? 1.0 : 0.0; });
    if (V != nullptr) {
        static_for<DP - 2>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            if (k < pad) return;
            const lds_f64* uk = refl + ((k - pad) * d - pad);
            double t = 0.0;
            static_for<DP - k - 1>([&](auto jj) {
                constexpr int j = k + 1 + decltype(jj)::value;
                t = __builtin_fma(z[j], uk[j], t);
            });
            t *= ihh[k];
            static_for<DP - k - 1>([&](auto jj) {
                constexpr int j = k + 1 + decltype(jj)::value;
                z[j] = __builtin_fma(-t, uk[j], z[j]);
            });
        });
    }
    GABO_EIGH_TICK(2);
    tridiag_ql_vectors<DP, 1>(dg, e, z);
    GABO_EIGH_TICK(3);
    wave_lds_order();
    if (V != nullptr && own) {
        static_for<DP>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (c >= pad) V[ra * d + (c - pad)] = z[c];
        });
    }
    static_for<DP>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (lane == c && c >= pad) A[(c - pad) * d + (c - pad)] = dg[c];
    });
    wave_lds_order();
}

// dispatch on the padded order; d in [kWaveEighMinDim, 32]
__device__ __forceinline__ void wave_eigh_any(double* A, double* V, double* bc, int d) {
    lds_f64* a = (lds_f64*)A;
    lds_f64* v = (lds_f64*)V;
    lds_f64* b = (lds_f64*)bc;
    if (d <= 8) wave_eigh<8>(a, v, b, d);
    else if (d <= 12) wave_eigh<12>(a, v, b, d);
    else if (d <= 16) wave_eigh<16>(a, v, b, d);
    else if (d <= 20) wave_eigh<20>(a, v, b, d);
    else if (d <= 24) wave_eigh<24>(a, v, b, d);
    else if (d <= 28) wave_eigh<28>(a, v, b, d);
    else wave_eigh<32>(a, v, b, d);
}

}  // namespace gabo
