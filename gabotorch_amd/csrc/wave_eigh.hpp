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
// Round 4: the reduction exchanges its column through v_readlane instead of LDS (wave_tridiagonalize), orders 9 ... 24 find the eigenpairs of T
// with one lane GROUP per eigenvalue (wave_eigh_rqi: multisection on Sturm counts + Rayleigh-quotient iteration + windowed Newton-Schulz
// step; QL is its fallback), and the QL path reverses T when its large end is on top.  DESIGN 4.9b, tools/ubench_eigh.hip.
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

// Householder reduction of the padded matrix to tridiagonal form, the first half of both solvers below.  Lane r keeps row r in a[];
// out: the tridiagonal (dg, e: wave-uniform), 1 / hh_k of the reflectors (ihh) and, when refl != null, the reflectors themselves
// (u_k, k >= pad, entry j at refl[(k - pad) d + (j - pad)]: at most (d - 2)(d - 1) doubles of a d x d LDS matrix).
template <int DP>
__device__ __forceinline__ void wave_tridiagonalize(lds_f64* A, lds_f64* refl, lds_f64* bc, int d, double (&dg)[DP], double (&e)[DP],
                                                    double (&ihh)[DP >= 3 ? DP - 2 : 1]) {
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
    static_for<DP - 2>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int n = DP - k - 1;                    // entries of column k below the diagonal
        if (k < pad) {                                   // wave-uniform: the identity block needs no reflector
            dg[k] = 1.0;
            e[k] = 0.0;
            ihh[k] = 0.0;
            return;
        }
#ifndef GABO_EIGH_LDS_REDUCTION
        // Round 4: no exchange through LDS.  Column k below the diagonal is ONE register pair spread over the lanes k+1 ... DP-1: v_readlane
        // with constant lane indices puts it into scalar registers, where every lane forms |x|^2 and the reflector redundantly (no
        // wave-wide sum) and uses the entries as the scalar operand of its FMAs; q comes back the same way.  One wave-wide sum per column
        // (u . p) is left.  Before: two sums and two LDS round trips per column, 27 k of the 75 k cycles at d = 20.
        const bool below = lane > k && lane < DP;
        const double x = below ? a[k] : 0.0;             // column k below the diagonal = entry k of the rows below (symmetry)
        double us[n];
        static_for<n>([&](auto ii) { us[decltype(ii)::value] = lane_value(a[k], k + 1 + decltype(ii)::value); });
        const double alpha = us[0];
        double n0 = 0.0, n1 = 0.0;
        static_for<n>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            if constexpr (i % 2 == 0) n0 = __builtin_fma(us[i], us[i], n0); else n1 = __builtin_fma(us[i], us[i], n1);
        });
        const double nn = n0 + n1;
        const double nrm = sqrt_pos(nn);
        const double hh = __builtin_fma(__builtin_fabs(alpha), nrm, nn);      // u = x + sign(x0)|x| e0, H = I - u u^T / hh
        const double inv_hh = hh == 0.0 ? 0.0 : rcp(hh);
        us[0] = alpha + copysign_d(nrm, alpha);
        const double u = (lane == k + 1) ? us[0] : x;
        if (refl != nullptr && below) refl[(k - pad) * d + (lane - pad)] = u;
        double p0 = 0.0, p1 = 0.0;
        static_for<n>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            if constexpr (i % 2 == 0) p0 = __builtin_fma(a[k + 1 + i], us[i], p0); else p1 = __builtin_fma(a[k + 1 + i], us[i], p1);
        });
        const double p = below ? (p0 + p1) * inv_hh : 0.0;
        const double kap = 0.5 * wave_allsum(u * p) * inv_hh;
        const double q = __builtin_fma(-kap, u, p);
        static_for<n>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const double qb = lane_value(q, k + 1 + i);
            a[k + 1 + i] = __builtin_fma(-q, us[i], __builtin_fma(-u, qb, a[k + 1 + i]));
        });
        dg[k] = lane_value(a[k], k);
        e[k] = hh == 0.0 ? alpha : -copysign_d(nrm, alpha);
        ihh[k] = inv_hh;
#else
        const bool below = lane > k && lane < DP;
        const double x = below ? a[k] : 0.0;             // column k below the diagonal = entry k of the rows below (symmetry)
        const double alpha = lane_value(a[k], k + 1);
        const double nn = wave_allsum(x * x);
        const double nrm = sqrt_pos(nn);
        const double hh = __builtin_fma(__builtin_fabs(alpha), nrm, nn);      // u = x + sign(x0)|x| e0, H = I - u u^T / hh
        const double inv_hh = hh == 0.0 ? 0.0 : rcp(hh);
        const double u = (lane == k + 1) ? alpha + copysign_d(nrm, alpha) : x;
        if (lane < DP) bc[2 * lane] = u;
        if (refl != nullptr && below) refl[(k - pad) * d + (lane - pad)] = u;
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
#endif
    });
    dg[DP - 2] = lane_value(a[DP - 2], DP - 2);
    e[DP - 2] = lane_value(a[DP - 2], DP - 1);
    dg[DP - 1] = lane_value(a[DP - 1], DP - 1);
    e[DP - 1] = 0.0;
}

// ---- round 4: one LANE GROUP per eigenvalue instead of the wave-serial QL recurrence ---------------------------------------------------------
// The QL phase of wave_eigh is one dependent chain (~19 instructions per rotation, ~1.7 d^2 / 2 rotations: 81 k of the 118 k cycles at d = 20)
// executed redundantly by every lane.  Here the d eigenpairs of the tridiagonal matrix T are found side by side:
//   0. T is scaled by a power of two (exactly) to |T| in [1/2, 1): the thresholds below are plain numbers and the Sturm sequence cannot overflow;
//   1. every eigenvalue gets m = 64 / d lanes; a few passes of multisection on the Sturm count of T - x I (each lane counts at its own abscissa:
//      one pass cuts every bracket (m + 1)-fold; the count is the number of sign changes of the three-term recurrence of the leading minors -
//      ONE dependent FMA per row, no division) put eigenvalue k into a bracket of <= 6e-5;
//   2. brackets that are not separated from their neighbours by their own width get up to 8 more passes; what is still crowded then (repeated
//      eigenvalues, clusters below 2e-9) sends the whole matrix to the QL path - one wave holds one matrix, so the decision is wave-uniform;
//   3. Rayleigh-quotient iteration per lane from the bracket's midpoint (pivoted LU of T - mu I with the forward substitution fused in, back
//      substitution, Rayleigh quotient).  A lane is done when the residual |T z - rq z| of its pair, computed explicitly, is below 2^-50
//      (floor of the fp64 evaluation: ~3e-16).  Typically 3 solves.  Safety net, independent of every count above: the brackets are disjoint (step 2), every quotient ends INSIDE its
//      own bracket and every pair has a rounding-level residual - d such pairs are all the eigenpairs; anything else falls back to QL;
//   4. each lane pushes ITS vector through the reflectors (wave-uniform LDS reads), the vectors go to V;
//   5. eigenvectors computed independently are orthogonal to eps / gap only, and a matrix FUNCTION sum f_k v_k v_k^T does not forgive that:
//      one Newton-Schulz step of the polar decomposition, V <- V (3 I - V^T V) / 2, restricted to the columns whose eigenvalues lie within
//      GABO_EIGH_RQI_ORTH_GAP of each other (a wave-uniform window of the sorted spectrum), squares the defect; two steps below a gap of 1e-6.
//      (The residual of a corrected vector stays at rounding level: the admixture of a neighbour is eps / gap, its residual against the
//      other eigenvalue is the gap.)
// Returns false when the QL path has to take over (V and the diagonal of A are untouched then; td holds T * scale, `unscale` = 1 / scale).
#ifndef GABO_EIGH_RQI_MIN_DP
#define GABO_EIGH_RQI_MIN_DP 8      /* with d >= 7 (below: the QL chain is as short as the fixed costs of the lane groups: 15.1 k against 15.4 k cycles at d = 5) */
#endif
#ifndef GABO_EIGH_RQI_MAX_DP
#define GABO_EIGH_RQI_MAX_DP 32
#endif
#ifndef GABO_EIGH_RQI_EXTRA_PASSES
#define GABO_EIGH_RQI_EXTRA_PASSES 1      /* one more pass (~1 k cycles) usually saves the fourth solve (~4 k): tools/ubench_eigh.hip */
#endif
#ifndef GABO_EIGH_RQI_ORTH_GAP
#define GABO_EIGH_RQI_ORTH_GAP 0.03
#endif
template <int DP>
__device__ __forceinline__ bool wave_eigh_rqi(lds_f64* A, lds_f64* V, const lds_f64* refl, lds_f64* td, int d, const double (&ihh)[DP >= 3 ? DP - 2 : 1],
                                              double (&ta)[DP], double (&tb)[DP], double& unscale) {
    // ta / tb: the tridiagonal matrix (diagonal, off-diagonal e_i between i and i + 1) in registers, wave-uniform; SCALED IN PLACE (the caller
    // undoes it with `unscale` when the QL path has to take over).  td: kWaveEighScratch doubles of LDS scratch (the sorted eigenvalues, step 5)
    const int lane = threadIdx.x & 63;
    const int pad = DP - d;                                // 0 ... 3 leading identity rows, decoupled (e[pad - 1] = 0): skipped below
    const int m = 64 / d;                                  // lanes per eigenvalue, 2 ... 7 (d >= 9)
    const int graw = lane / m;
    const bool spare = graw >= d;                          // lanes beyond the last group compute along with it; their votes are outside every group's mask
    const int grp = spare ? d - 1 : graw;
    const int j = spare ? 0 : lane - grp * m;
    const int leader = grp * m;
    unscale = 1.0;
    // ---- Gershgorin bracket of the active block
    double gl = 1e300, gu = -1e300;
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const double off = __builtin_fabs(tb[i]) + (i > 0 ? __builtin_fabs(tb[i > 0 ? i - 1 : 0]) : 0.0);     // (e[pad - 1] = 0, e[DP - 1] = 0)
        const double di = ta[i];
        double lo_i = di - off, hi_i = di + off;
        if constexpr (i < 3) {                            // (the identity rows of the padding do not take part; pad <= 3 here)
            lo_i = i < pad ? 1e300 : lo_i;
            hi_i = i < pad ? -1e300 : hi_i;
        }
        gl = __builtin_fmin(gl, lo_i);
        gu = __builtin_fmax(gu, hi_i);
    });
    const double tnorm = __builtin_fmax(__builtin_fabs(gl), __builtin_fabs(gu));
    if (!(tnorm > 1e-290) || !(tnorm < 1e290)) return false;         // zero matrix, NaN, extreme scales: the QL path
    int ex;
    (void)frexp(tnorm, &ex);
    const double scale = ldexp(1.0, -ex);
    unscale = ldexp(1.0, ex);
    // (the identity rows in front of the matrix become eigenvalues 4 above the scaled spectrum: the loops below run over all DP rows without a
    // branch, the counts below x < 1 and the vectors - zero in those rows - do not see them)
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        ta[i] *= scale;
        tb[i] *= scale;                                   // (zero in the padding rows already)
        if constexpr (i < 3) ta[i] = i < pad ? 4.0 : ta[i];
    });
    double tb2[DP];
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        tb2[i] = (i == 0) ? 0.0 : -(tb[i > 0 ? i - 1 : 0] * tb[i > 0 ? i - 1 : 0]);
    });
    double lo = gl * scale - 2e-15, hi = gu * scale + 2e-15;
    const double inv = 1.0 / (double)(m + 1);
    const double frac = (double)(j + 1) * inv;
    const unsigned long long gmask = (1ull << m) - 1ull;
    const int passes = (m == 2 ? 9 : (m == 3 ? 7 : (m <= 5 ? 6 : 5))) + GABO_EIGH_RQI_EXTRA_PASSES;       // (m + 1)^passes >= 15625
    const int nxt = (grp + 1 < d ? grp + 1 : grp) * m;
    bool isolated = false;
    int npass = 0, nsolve = 0;        // (instrumentation: tools/ubench_eigh.hip)
    // (no bracket narrower than ~2e-9: eigenvalues closer than that are left to the QL path - their computed vectors mix by residual / gap,
    // more than the two Newton-Schulz steps of step 5 take out)
    const int pass_cap = m == 2 ? 19 : (m == 3 ? 15 : (m == 4 ? 13 : (m <= 6 ? 11 : 10)));
    const int pass_end = passes + 8 < pass_cap ? passes + 8 : pass_cap;
    for (int pass = 0; pass < pass_end; ++pass) {
        ++npass;
        const double w = hi - lo;
        const double x = __builtin_fma(w, frac, lo);
        // sign changes of p_0 = 1, p_i = (a_i - x) p_(i-1) - e_(i-1)^2 p_(i-2) = eigenvalues below x; |p_i| <= 5^i.  The signs are shifted
        // into one word, newest at bit 0: four instructions per row.  (A p_i that is exactly zero should take the sign opposite to its
        // predecessor's; it takes the sign of its zero instead - a count that is off by one misplaces a bracket, which step 3's safety net
        // catches.)
        double pm2 = 1.0, pm1 = 1.0;
        unsigned signs = 0u;
        static_for<DP>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const double p = __builtin_fma(ta[i] - x, pm1, tb2[i] * pm2);
            signs = __builtin_amdgcn_alignbit(signs, (unsigned)__double2hiint(p), 31);
            pm2 = pm1;
            pm1 = p;
        });
        const int cnt = __builtin_popcount(signs ^ (signs >> 1));       // (bit DP would be the sign of p_0: 0)
        const unsigned long long above = __builtin_amdgcn_ballot_w64(cnt >= grp + 1);      // x lies above eigenvalue `grp` (0-based, ascending)
        const unsigned mine = (unsigned)((above >> leader) & gmask);
        const int first = mine ? __builtin_ctz(mine) : m;
        const double nlo = first == 0 ? lo : __builtin_fma(w, (double)first * inv, lo);
        const double nhi = first == m ? hi : __builtin_fma(w, (double)(first + 1) * inv, lo);
        lo = nlo;
        hi = nhi;
        if (pass + 1 >= passes) {
            const double lo_next = __shfl(lo, nxt, 64);
            const bool crowded = !spare && grp + 1 < d && !(lo_next - hi >= hi - lo);
            if (__builtin_amdgcn_ballot_w64(crowded) == 0) { isolated = true; break; }
        }
    }
    GABO_EIGH_TICK(4);
    if (!isolated) return false;
    // ---- Rayleigh-quotient iteration on T (every lane for the eigenvalue of its group)
    constexpr double tiny = 1.2e-16;
    double z[DP];
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        z[i] = (i < pad) ? 0.0 : 1.0 + 0.25 * (double)((i * 7) % 5);
    });
    double mu = 0.5 * (lo + hi), rho = mu;
    bool done = false, conv = false;
    for (int it = 0; it < 7; ++it) {
        ++nsolve;
        // pivoted LU of T - mu I, the forward substitution applied to z on the way (rows of U: 1 / u0, u1, [e_(k+1) where rows were interchanged])
        double iu0[DP], u1[DP];
        double cu = ta[0] - mu, cv = tb[0];
        static_for<DP - 1>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const double ck = tb[k];
            const double an = ta[k + 1] - mu, bn = tb[k + 1];         // (e[DP - 1] = 0)
            const bool swp = __builtin_fabs(ck) > __builtin_fabs(cu);
            double piv = swp ? ck : cu;
            const double other = swp ? cu : ck;
            piv = (__builtin_fabs(piv) < tiny) ? copysign_d(tiny, piv) : piv;
            const double ip = rcp(piv);
            const double mult = other * ip;
            const double ux = swp ? an : cv, uy = swp ? cv : an;      // the pivot row's superdiagonal entry and the entry below it
            // (the interchange flag rides in the last mantissa bit of the stored reciprocal - a perturbation of U by half an ulp - instead of
            // a register of its own per row)
            iu0[k] = __hiloint2double(__double2hiint(ip), (__double2loint(ip) & ~1) | (swp ? 1 : 0));
            u1[k] = ux;
            cu = __builtin_fma(-mult, ux, uy);
            cv = (swp ? -mult : 1.0) * bn;
            const double yk = z[k], yn = z[k + 1];
            const double zx = swp ? yn : yk, zy = swp ? yk : yn;
            z[k] = zx;
            z[k + 1] = __builtin_fma(-mult, zx, zy);
        });
        cu = (__builtin_fabs(cu) < tiny) ? copysign_d(tiny, cu) : cu;
        iu0[DP - 1] = rcp(cu);      // (no row below: its flag is never read)
        static_for_down<DP - 1, 0>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            double acc = z[k];
            if constexpr (k + 1 < DP) acc = __builtin_fma(-u1[k + 1 < DP ? k : 0], z[k + 1 < DP ? k + 1 : k], acc);
            if constexpr (k + 2 < DP) acc = __builtin_fma((__double2loint(iu0[k]) & 1) ? -tb[k + 1 < DP ? k + 1 : k] : 0.0, z[k + 2 < DP ? k + 2 : k], acc);
            z[k] = acc * iu0[k];
        });
        double nn = 0.0, big = 0.0;
        static_for<DP>([&](auto ii) { big = __builtin_fmax(big, __builtin_fabs(z[decltype(ii)::value])); });
        const double sc = rcp(big);                                                      // (against overflow of the squares)
        static_for<DP>([&](auto ii) { z[decltype(ii)::value] *= sc; nn = __builtin_fma(z[decltype(ii)::value], z[decltype(ii)::value], nn); });
        const double inr = rsqrt_nz(nn);
        static_for<DP>([&](auto ii) { z[decltype(ii)::value] *= inr; });
        // Rayleigh quotient of the normalised vector - the eigenvalue, and the next shift while it stays with its bracket - and the residual
        // |T z - rq z| of the pair: a lane is done when that is at rounding level (|T| < 1)
        double tz[DP];
        double rq0 = 0.0, rq1 = 0.0;
        static_for<DP>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            double acc = ta[i] * z[i];
            if constexpr (i > 0) acc = __builtin_fma(tb[i > 0 ? i - 1 : 0], z[i > 0 ? i - 1 : 0], acc);
            if constexpr (i + 1 < DP) acc = __builtin_fma(tb[i], z[i + 1 < DP ? i + 1 : i], acc);
            tz[i] = acc;
            if constexpr (i % 2 == 0) rq0 = __builtin_fma(acc, z[i], rq0); else rq1 = __builtin_fma(acc, z[i], rq1);
        });
        const double rq = rq0 + rq1;
        double rs0 = 0.0, rs1 = 0.0;
        static_for<DP>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const double ri = __builtin_fma(-rq, z[i], tz[i]);
            if constexpr (i % 2 == 0) rs0 = __builtin_fma(ri, ri, rs0); else rs1 = __builtin_fma(ri, ri, rs1);
        });
        done = done || (rs0 + rs1 <= 0x1p-100);
        rho = rq;
        mu = (rho >= lo && rho <= hi) ? rho : mu;
        if (__builtin_amdgcn_ballot_w64(!done && !spare) == 0) { conv = true; break; }
    }
    GABO_EIGH_TICK(5);
    if (!conv || __builtin_amdgcn_ballot_w64(!spare && !(rho >= lo && rho <= hi)) != 0) return false;
    // ---- back to the original basis: this lane's vector through H_pad ... H_(DP-3), last reflector first
    static_for_down<DP - 3, 0>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        if constexpr (k < 3) { if (k < pad) return; }
        const lds_f64* uk = refl + ((k - pad) * d - pad);
        double u[DP - k - 1];
        static_for<DP - k - 1>([&](auto jj) { u[decltype(jj)::value] = uk[k + 1 + decltype(jj)::value]; });
        __builtin_amdgcn_sched_barrier(0);
        double t0 = 0.0, t1 = 0.0;
        static_for<DP - k - 1>([&](auto jj) {
            constexpr int q = decltype(jj)::value;
            if constexpr (q % 2 == 0) t0 = __builtin_fma(z[k + 1 + q], u[q], t0); else t1 = __builtin_fma(z[k + 1 + q], u[q], t1);
        });
        const double t = (t0 + t1) * ihh[k];
        static_for<DP - k - 1>([&](auto jj) {
            constexpr int q = decltype(jj)::value;
            z[k + 1 + q] = __builtin_fma(-t, u[q], z[k + 1 + q]);
        });
    });
    GABO_EIGH_TICK(6);
    wave_lds_order();                                       // every lane is done with the reflectors: V may be overwritten
    const bool writer = !spare && j == 0;
    if (writer) {
        static_for<DP>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (c >= pad) V[(c - pad) * d + grp] = z[c];
        });
        A[grp * d + grp] = rho * unscale;
    }
    wave_lds_order();
    // ---- Newton-Schulz step(s) against the columns whose eigenvalues are close: a window of the sorted spectrum, the same for every group
    // (the sorted eigenvalues go through the scratch that held T; the spectrum is ascending, so the largest offset at which ANY group still
    // has a neighbour within the threshold is the window.  Written without a run-time loop: the loop form with __shfl and an early exit
    // took InstCombine four minutes per instantiation)
    if (writer) td[grp] = rho;
    wave_lds_order();
    int win = 0;
    double gap_min = 1.0;
    static_for<DP - 1>([&](auto kk) {
        constexpr int k = decltype(kk)::value + 1;
        const bool has = !spare && grp + k < d;
        const double other = td[has ? grp + k : grp];
        const bool near = has && other - rho < GABO_EIGH_RQI_ORTH_GAP;
        if constexpr (k == 1) gap_min = has ? other - rho : 1.0;
        win = __builtin_amdgcn_ballot_w64(near) != 0 ? k : win;
    });
#ifdef GABO_EIGH_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x == 0) gabo_eigh_clk[7] = npass * 10000 + nsolve * 100 + win;
#endif
    const int ns_steps = win == 0 ? 0 : (__builtin_amdgcn_ballot_w64(gap_min < 1e-6) != 0 ? 2 : 1);
    for (int ns = 0; ns < ns_steps; ++ns) {
        double zn[DP];
        {
            double g0 = 0.0, g1 = 0.0;
            static_for<DP>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                if constexpr (r % 2 == 0) g0 = __builtin_fma(z[r], z[r], g0); else g1 = __builtin_fma(z[r], z[r], g1);
            });
            const double cf = 0.5 * (3.0 - (g0 + g1));
            static_for<DP>([&](auto rr) { zn[decltype(rr)::value] = cf * z[decltype(rr)::value]; });
        }
        for (int o = -win; o <= win; ++o) {
            if (o == 0) continue;
            const int col = grp + o;
            const bool ok = col >= 0 && col < d;
            const lds_f64* vc = V + (ok ? col : grp);
            double cv[DP];
            static_for<DP>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                cv[r] = 0.0;
                if constexpr (r < 3) { if (r < pad) return; }
                cv[r] = vc[(r - pad) * d];
            });
            __builtin_amdgcn_sched_barrier(0);      // (all reads of the column in flight before the first use: the scheduler would pair each with its FMA)
            double g0 = 0.0, g1 = 0.0;
            static_for<DP>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                if constexpr (r % 2 == 0) g0 = __builtin_fma(z[r], cv[r], g0); else g1 = __builtin_fma(z[r], cv[r], g1);
            });
            const double cf = ok ? -0.5 * (g0 + g1) : 0.0;
            static_for<DP>([&](auto rr) { zn[decltype(rr)::value] = __builtin_fma(cf, cv[decltype(rr)::value], zn[decltype(rr)::value]); });
        }
        static_for<DP>([&](auto rr) { z[decltype(rr)::value] = zn[decltype(rr)::value]; });
        wave_lds_order();                                   // all reads of the old columns are done
        if (writer) {
            static_for<DP>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (c >= pad) V[(c - pad) * d + grp] = z[c];
            });
        }
        wave_lds_order();
    }
    return true;
}

// A (d x d, symmetric, row-major, LDS): on return its DIAGONAL holds the eigenvalues (unordered; the rest of A is left as it was);
// V (d x d, LDS, may be null: eigenvalues only): eigenvectors in columns; bc: kWaveEighScratch doubles of LDS.  DP - 3 <= d <= DP (DP = 8: 2 <= d <= 8).
// Called by all 64 lanes of one wave (any other waves of the block wait at the caller's barrier).
// LANE_GROUPS = false: the QL path only (orders up to 6: the lane-group code would only cost them its callee-saved registers)
template <int DP, bool LANE_GROUPS = true>
__device__ __attribute__((noinline)) void wave_eigh(lds_f64* A, lds_f64* V, lds_f64* bc, int d_arg) {
    const int d = __builtin_amdgcn_readfirstlane(d_arg);      // (arguments of a function arrive in vector registers: every branch on the padding would be an exec-mask branch)
    const int lane = threadIdx.x & 63;
    const int pad = DP - d;
    const int ra = lane - pad;
    const bool own = ra >= 0 && lane < DP;
    GABO_EIGH_TICK(0);
    double dg[DP], e[DP], ihh[DP >= 3 ? DP - 2 : 1];
    lds_f64* refl = V;                                    // overwritten by Z at the end
    wave_tridiagonalize<DP>(A, refl, bc, d, dg, e, ihh);
    GABO_EIGH_TICK(1);
#ifndef GABO_EIGH_NO_RQI
    // (d = 7, 8: 21.4 k -> 18.9 k, 25.2 k -> 20.6 k cycles; d <= 6: no gain, the QL path stays - tools/ubench_eigh.hip)
    if constexpr (LANE_GROUPS && DP >= GABO_EIGH_RQI_MIN_DP && DP <= GABO_EIGH_RQI_MAX_DP) {
        if (V != nullptr && (DP > 8 || d >= 7)) {
            // T stays in the registers it is in (scaled in place by the lane-group solver: a power of two, exact both ways)
            double unscale;
            if (wave_eigh_rqi<DP>(A, V, refl, bc, d, ihh, dg, e, unscale)) {
                GABO_EIGH_TICK(2);
                GABO_EIGH_TICK(3);
                return;
            }
            static_for<DP>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                dg[c] = c < pad ? 1.0 : dg[c] * unscale;
                e[c] = c < pad ? 0.0 : e[c] * unscale;
            });
        }
    }
#endif
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
    // The QL iteration deflates from the top of T and wants the SMALL end of a graded matrix there (LAPACK's dsteqr picks QL or QR by the same
    // comparison of the two ends); with the large end on top, a cluster of small eigenvalues exhausts the 60 iterations of a stage
    // (tests/test_gpu_manifold_ops.py::test_lane_group_eigen_solver_on_many_spectra, graded matrices: 5e-10 |A| before).  T is reversed
    // then - J T J, the columns of Q with it - and the result reversed back.
    constexpr int kMaxPad = DP == 8 ? 6 : 3;             // (wave_eigh_any: the smallest padded order; orders 2 ... 4 come from spd_backward2.hip)
    double d_top = dg[0];
    static_for<kMaxPad>([&](auto ii) { d_top = (pad == decltype(ii)::value + 1) ? dg[decltype(ii)::value + 1] : d_top; });
    const bool reversed = __builtin_fabs(d_top) > __builtin_fabs(dg[DP - 1]);
    // (only the active block is reversed: the identity rows of the padding stay in front, where their stages deflate at once)
    auto flip = [&](auto pp, bool with_e) {
        constexpr int P = decltype(pp)::value;
        static_for<(DP - P) / 2>([&](auto ii) {
            constexpr int i = P + decltype(ii)::value, j = DP - 1 - decltype(ii)::value;
            const double t0 = dg[i]; dg[i] = dg[j]; dg[j] = t0;
            const double t1 = z[i]; z[i] = z[j]; z[j] = t1;
        });
        if (with_e) {
            static_for<(DP - 1 - P) / 2>([&](auto ii) {
                constexpr int i = P + decltype(ii)::value, j = DP - 2 - decltype(ii)::value;
                const double t0 = e[i]; e[i] = e[j]; e[j] = t0;
            });
        }
    };
    auto flip_any = [&](bool with_e) {
        static_for<kMaxPad + 1>([&](auto pp) { if (pad == decltype(pp)::value) flip(pp, with_e); });
    };
    if (reversed) flip_any(true);
    tridiag_ql_vectors<DP, 1>(dg, e, z);
    if (reversed) flip_any(false);
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

// Sum over each half of the wave (lanes 0..31, lanes 32..63), the half's total in every lane of that half.  All 64 lanes active.
__device__ __forceinline__ double half_wave_allsum(double v) {
    v += dpp_fetch<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_fetch<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_fetch<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_fetch<0x140, 0xf>(v);       // row_mirror: every lane of a row of 16 holds the row's sum
    v += dpp_fetch<0x142, 0xa>(v);       // row_bcast15 into rows 1 and 3: they now hold the sums of their halves
    const double lo = lane_value(v, 31), hi = lane_value(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}

// The two EXTREME eigenpairs of one d x d symmetric matrix by one wave, for callers that bound lambda_max / lambda_min and differentiate
// them (eigenvalue constraints stated in the original space of a nested SPD mapping: nested_spd_constraints_utils.py:14-73).  Householder
// reduction as above, then - instead of the QL iteration over the whole spectrum, 70 % of wave_eigh -
//   * both eigenvalues by multisection on the Sturm count of T - x I: lanes 0..31 work on lambda_max and lanes 32..63 on lambda_min, each lane
//     counts at its own abscissa, so one pass of the d-step recurrence cuts both brackets 33-fold (12 passes: 2^-60 of the Gershgorin width);
//   * both eigenvectors of T by three steps of inverse iteration on the pivoted LU factors of T - lambda I (each half of the wave for its
//     own lambda, the recurrences run redundantly in the lanes of the half);
//   * the reflectors applied to both vectors at once (lane = (half, component), dot products by a half-wave DPP sum).
// A: in: the matrix; out: A[0..d) = eigenvector of lambda_max, A[d..2d) = eigenvector of lambda_min (unit norm, sign arbitrary).
// W: d x d of LDS scratch (the reflectors).  bc: kWaveEighScratch doubles; out: bc[0] = lambda_max, bc[1] = lambda_min.  DP - 3 <= d <= DP.
template <int DP>
__device__ __attribute__((noinline)) void wave_eig_extremes(lds_f64* A, lds_f64* W, lds_f64* bc, int d_arg) {
    const int d = __builtin_amdgcn_readfirstlane(d_arg);      // (see wave_eigh)
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, r = lane & 31;
    const int pad = DP - d;
    double dg[DP], e[DP], ihh[DP >= 3 ? DP - 2 : 1];
    wave_tridiagonalize<DP>(A, W, bc, d, dg, e, ihh);
    // ---- Gershgorin bracket of the spectrum of the active block
    double gl = 0.0, gu = 0.0;
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        if (i < pad) return;
        const double off = __builtin_fabs(e[i]) + (i > 0 ? __builtin_fabs(e[i > 0 ? i - 1 : 0]) : 0.0);     // (e[pad - 1] = 0)
        const double lo_i = dg[i] - off, hi_i = dg[i] + off;
        gl = (i == pad || lo_i < gl) ? lo_i : gl;
        gu = (i == pad || hi_i > gu) ? hi_i : gu;
    });
    const double tnorm = __builtin_fmax(__builtin_fabs(gl), __builtin_fabs(gu));
    // Sturm counts as in wave_eigh_rqi (round 4): T scaled by a power of two to |T| in [1/2, 1), the count = sign changes of the minors'
    // three-term recurrence - one dependent FMA per row, no division (a corrected reciprocal per row before: 12 instructions instead of 4);
    // the identity rows of the padding become eigenvalues 4, above every abscissa
    int ex = 0;
    double scale = 1.0, unscale = 1.0;
    if (tnorm > 0.0 && tnorm < 1e300) {
        (void)frexp(tnorm, &ex);
        scale = ldexp(1.0, -ex);
        unscale = ldexp(1.0, ex);
    }
    double ta[DP], tb2[DP];
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        ta[i] = i < pad ? 4.0 : dg[i] * scale;
        const double es = (i == 0) ? 0.0 : e[i > 0 ? i - 1 : 0] * scale;
        tb2[i] = -(es * es);
    });
    double lo = gl * scale - 2e-15, hi = gu * scale + 2e-15;         // per half: the bracket of its eigenvalue
    const double frac = (double)(r + 1) * (1.0 / 33.0);
    const int target = half ? 1 : d;                     // count(x) >= target  <=>  x is above the eigenvalue this half looks for
    for (int pass = 0; pass < 12; ++pass) {
        const double w = hi - lo;
        const double x = __builtin_fma(w, frac, lo);
        double pm2 = 1.0, pm1 = 1.0;
        unsigned signs = 0u;
        static_for<DP>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const double p = __builtin_fma(ta[i] - x, pm1, tb2[i] * pm2);
            signs = __builtin_amdgcn_alignbit(signs, (unsigned)__double2hiint(p), 31);
            pm2 = pm1;
            pm1 = p;
        });
        const int cnt = __builtin_popcount(signs ^ (signs >> 1));
        const unsigned long long above = __builtin_amdgcn_ballot_w64(cnt >= target);
        const unsigned mine = half ? (unsigned)(above >> 32) : (unsigned)above;
        const int first = mine ? __builtin_ctz(mine) : 32;                               // first abscissa above the eigenvalue
        const double nlo = first == 0 ? lo : __builtin_fma(w, (double)first * (1.0 / 33.0), lo);
        const double nhi = first == 32 ? hi : __builtin_fma(w, (double)(first + 1) * (1.0 / 33.0), lo);
        lo = nlo;
        hi = nhi;
    }
    const double lam = 0.5 * (lo + hi) * unscale;
    // ---- pivoted LU of T - lam I (rows of U: 1 / u0, u1, u2; multipliers ml; interchanges sw)
    const double tiny = __builtin_fmax(2.3e-16 * tnorm, 1e-300);
    double iu0[DP], u1[DP], ml[DP];                       // (the second super-diagonal of U is e[k + 1] where rows were interchanged, 0 elsewhere)
    bool sw[DP];
    double cu = dg[0] - lam, cv = e[0];
    static_for<DP - 1>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const double ck = e[k];
        const double an = dg[k + 1] - lam, bn = (k + 1 < DP - 1) ? e[k + 1 < DP - 1 ? k + 1 : 0] : 0.0;
        const bool swp = __builtin_fabs(ck) > __builtin_fabs(cu);
        double piv = swp ? ck : cu;
        const double other = swp ? cu : ck;
        piv = (__builtin_fabs(piv) < tiny) ? copysign_d(tiny, piv) : piv;
        const double ip = rcp(piv);
        const double mult = other * ip;
        iu0[k] = ip;
        u1[k] = swp ? an : cv;
        ml[k] = mult;
        sw[k] = swp;
        const double ncu = swp ? __builtin_fma(-mult, an, cv) : __builtin_fma(-mult, cv, an);
        cv = swp ? -mult * bn : bn;
        cu = ncu;
    });
    cu = (__builtin_fabs(cu) < tiny) ? copysign_d(tiny, cu) : cu;
    iu0[DP - 1] = rcp(cu);
    u1[DP - 1] = 0.0;
    sw[DP - 1] = false;
    // ---- inverse iteration: U z = b first (a start vector with the L-solve already "applied"), then two full solves
    double z[DP];
    static_for<DP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        z[i] = (i < pad) ? 0.0 : 1.0 + 0.25 * (double)((i * 7) % 5);
    });
    for (int it = 0; it < 3; ++it) {
        if (it > 0) {
            static_for<DP - 1>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                const double yk = z[k], yn = z[k + 1];
                z[k] = sw[k] ? yn : yk;
                z[k + 1] = sw[k] ? __builtin_fma(-ml[k], yn, yk) : __builtin_fma(-ml[k], yk, yn);
            });
        }
        static_for_down<DP - 1, 0>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            double acc = z[k];
            if constexpr (k + 1 < DP) acc = __builtin_fma(-u1[k], z[k + 1 < DP ? k + 1 : k], acc);
            if constexpr (k + 2 < DP) acc = __builtin_fma(sw[k] ? -e[k + 1 < DP ? k + 1 : k] : 0.0, z[k + 2 < DP ? k + 2 : k], acc);
            z[k] = acc * iu0[k];
        });
        double nn = 0.0, big = 0.0;
        static_for<DP>([&](auto ii) { big = __builtin_fmax(big, __builtin_fabs(z[decltype(ii)::value])); });
        const double sc = rcp(big);                                                      // (against overflow of the squares)
        static_for<DP>([&](auto ii) { z[decltype(ii)::value] *= sc; nn = __builtin_fma(z[decltype(ii)::value], z[decltype(ii)::value], nn); });
        const double inr = rsqrt_nz(nn);
        static_for<DP>([&](auto ii) { z[decltype(ii)::value] *= inr; });
    }
    // ---- back to the original basis: y = H_pad ... H_{DP-3} z, component r of this half's vector in lane (half, r)
    double zr = 0.0;
    static_for<DP>([&](auto cc) { zr = (r == decltype(cc)::value) ? z[decltype(cc)::value] : zr; });
    static_for_down<DP - 3, 0>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        if (k < pad) return;
        const double u = (r > k && r < DP) ? W[(k - pad) * d + (r - pad)] : 0.0;
        const double dot = half_wave_allsum(u * zr) * ihh[k];
        zr = __builtin_fma(-dot, u, zr);
    });
    wave_lds_order();
    if (r >= pad && r < DP) A[half * d + (r - pad)] = zr;
    if (r == 0) bc[half] = lam;
    wave_lds_order();
}

// MAXDP: the largest padded order the caller links (a kernel that can reach an instantiation is allocated for its registers: the
// single-launch trust-region solve stops at 24)
template <int MAXDP = 32>
__device__ __forceinline__ void wave_eig_extremes_any(double* A, double* W, double* bc, int d) {
    lds_f64* a = (lds_f64*)A;
    lds_f64* w = (lds_f64*)W;
    lds_f64* b = (lds_f64*)bc;
    if (d <= 8) wave_eig_extremes<8>(a, w, b, d);
    else if (d <= 12) wave_eig_extremes<12>(a, w, b, d);
    else if (d <= 16) wave_eig_extremes<16>(a, w, b, d);
    else if (d <= 20) wave_eig_extremes<20>(a, w, b, d);
    else if (d <= 24 || MAXDP <= 24) wave_eig_extremes<24>(a, w, b, d);
    else if constexpr (MAXDP > 24) {
        if (d <= 28) wave_eig_extremes<28>(a, w, b, d);
        else wave_eig_extremes<32>(a, w, b, d);
    }
}

// dispatch on the padded order; d in [kWaveEighMinDim, 32]
__device__ __forceinline__ void wave_eigh_any(double* A, double* V, double* bc, int d) {
    lds_f64* a = (lds_f64*)A;
    lds_f64* v = (lds_f64*)V;
    lds_f64* b = (lds_f64*)bc;
    if (d <= 6) wave_eigh<8, false>(a, v, b, d);
    else if (d <= 8) wave_eigh<8>(a, v, b, d);
    else if (d <= 12) wave_eigh<12>(a, v, b, d);
    else if (d <= 16) wave_eigh<16>(a, v, b, d);
    else if (d <= 20) wave_eigh<20>(a, v, b, d);
    else if (d <= 24) wave_eigh<24>(a, v, b, d);
    else if (d <= 28) wave_eigh<28>(a, v, b, d);
    else wave_eigh<32>(a, v, b, d);
}

}  // namespace gabo
