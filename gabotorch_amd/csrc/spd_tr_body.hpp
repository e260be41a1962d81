// Kernel templates of the trust-region iteration (see spd_tr.hip for the description).
#pragma once
#include "spd_acq_body.hpp"
#include "spd_tcg_body.hpp"
#include "nested_spd_lift.hpp"

namespace gabo {

struct TrWs {
    TcgWs tcg;
    double *x_fd, *eg_fd, *eg_fd0, *val_fd, *xp_mandel, *eg_prop, *fx_prop, *rhoden, *xp_mat, *F;
    size_t bytes;
};

static __host__ __device__ inline TrWs tr_layout(void* base, int64_t R, int d, int C, int64_t n) {
    TrWs t;
    t.tcg = tcg_layout(base, R, d, C);
    size_t off = (t.tcg.bytes + 15) & ~(size_t)15;
    double* p = (double*)((char*)base + off);
    const int64_t dv = (int64_t)d * (d + 1) / 2;
    t.x_fd = p;       p += R * dv;
    t.eg_fd = p;      p += R * dv;
    t.eg_fd0 = p;     p += R * dv;      // gradient at the FIRST FD point of the current x (reused while x does not move)
    t.val_fd = p;     p += R;
    t.xp_mandel = p;  p += R * dv;
    t.eg_prop = p;    p += R * dv;
    t.fx_prop = p;    p += R;
    t.rhoden = p;     p += R;
    t.xp_mat = p;     p += R * (int64_t)d * d;
    t.F = p;          p += R * dv * n;
    t.bytes = (size_t)((char*)p - (char*)base);
    return t;
}

// The constraints the library can evaluate itself (no host callable needed): extreme eigenvalues of the iterate
// (max/min_eigenvalue_constraint_torch, spd_constraints_utils_torch.py:17-50) - kind 0: bound - lambda_max(x) >= 0, 1: lambda_min(x) - bound >= 0 -
// and of the iterate LIFTED to the original space of a nested SPD mapping (max/min_eigenvalue_nested_spd_constraint,
// nested_spd_constraints_utils.py:14-73) - kinds 2 / 3, with lift_w, lift_p (big_dim x d) and lift_x0 (big_dim x big_dim) from
// gabo_nested_spd_lift_prepare; the wave evaluates those through nested_extremes_body in `nlds` (dynamic LDS).
constexpr int kTrNestedMaxDim = 24;      // largest original dimension of the nested kinds inside the solve kernel (gabo_spd_tr_solve checks it)
struct BuiltinCons {
    int n;
    int strict;
    int kind[kMaxCons];
    double bound[kMaxCons];
    int big_dim;
    const double *lift_w, *lift_p, *lift_x0;
};

static __host__ __device__ inline bool builtin_has_kind(const BuiltinCons& B, bool nested) {
#ifdef GABO_TR_NO_NESTED    /* A/B: the solve kernels without the nested kinds (their eigen-solver of order 24 sets the register budget) */
    if (nested) return false;
#endif
    for (int k = 0; k < B.n; ++k)
        if ((B.kind[k] >= 2) == nested) return true;
    return false;
}

// eigen-decomposition of the symmetric D x D matrix at `a` (row-major, global or LDS), every lane redundantly (D <= 8)
template <int D>
__device__ __forceinline__ void eig_extremes(const double* __restrict__ a, double (&lam)[D], double (&v)[D * D]) {
    constexpr int T = tri_size(D);
    double m[T];
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) { constexpr int c = decltype(cc)::value; m[tri(r, c)] = 0.5 * (a[r * D + c] + a[c * D + r]); });
    });
    sym_eig_reg<D>(m, lam, v);
}

// Eigenvector z (unit length) of the symmetric tridiagonal matrix (dg, e) for its eigenvalue lam by the twisted factorisation (Parlett & Dhillon;
// LAPACK dlar1v): pivots of the LDL^T factorisation of T - lam I from the top (dp) and of the UDU^T factorisation from the bottom (dm) meet at the
// index k where gamma_k = dp_k + dm_k - (d_k - lam) is smallest in magnitude; z_k = 1 and the two recurrences run outwards from there.  All operands
// are wave-uniform here (every lane holds the same matrix), so the twist index is made a scalar and its D cases are scalar branches.  Zero pivots
// are floored at 1e-3 eps of the matrix scale (an exactly decoupled block then gives the unit vector it should).
template <int D>
__device__ __forceinline__ void tridiag_twisted_vector(const double (&dg)[D], const double (&e)[D], double lam, double (&z)[D]) {
    double scale = __builtin_fabs(lam);
    static_for<D>([&](auto kk) { scale = __builtin_fmax(scale, __builtin_fabs(dg[decltype(kk)::value])); });
    const double floor_p = __builtin_fmax(scale * 2.2e-19, 1e-290);
    auto safe = [&](double p) { return __builtin_fabs(p) < floor_p ? copysign_d(floor_p, p) : p; };
    double dp[D], dm[D], lp[D >= 2 ? D - 1 : 1], um[D >= 2 ? D - 1 : 1];
    dp[0] = safe(dg[0] - lam);
    static_for<D - 1>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        lp[i] = e[i] * rcp(dp[i]);
        dp[i + 1] = safe((dg[i + 1] - lam) - lp[i] * e[i]);
    });
    dm[D - 1] = safe(dg[D - 1] - lam);
    static_for_down<D - 2, 0>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        um[i] = e[i] * rcp(dm[i + 1]);
        dm[i] = safe((dg[i] - lam) - um[i] * e[i]);
    });
    double gbest = __builtin_inf();
    int kbest = 0;
    static_for<D>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const double gam = __builtin_fabs((dp[k] + dm[k]) - (dg[k] - lam));
        const bool better = gam < gbest;
        gbest = better ? gam : gbest;
        kbest = better ? k : kbest;
    });
    kbest = __builtin_amdgcn_readfirstlane(kbest);
    static_for<D>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        if (kbest == K) {
            z[K] = 1.0;
            static_for_down<K - 1, 0>([&](auto ii) { constexpr int i = decltype(ii)::value; z[i] = -lp[i] * z[i + 1]; });
            static_for<D - 1 - K>([&](auto ii) { constexpr int i = K + decltype(ii)::value; z[i + 1] = -um[i] * z[i]; });
        }
    });
    double nn = 0.0;
    static_for<D>([&](auto kk) { nn = __builtin_fma(z[decltype(kk)::value], z[decltype(kk)::value], nn); });
    const double inv = rsqrt_nz(nn);
    static_for<D>([&](auto kk) { z[decltype(kk)::value] *= inv; });
}

// lambda_max / lambda_min of the symmetric D x D matrix at `a` and, for each that is asked for, a unit eigenvector - every lane redundantly.
// The eigenvalues by the eigenvalue-only QL recurrence (the bits sym_eig_reg gives), ONE vector per extreme by the twisted factorisation of the
// tridiagonal form, carried back through the Householder reflectors.  The strict variant's feasibility test uses the eigenvalues alone; for the
// constraint gradients it is an opt-in experiment (-DGABO_BUILTIN_TWISTED, see builtin_constraints).
template <int D>
__device__ __forceinline__ void eig_extreme_pairs(const double* __restrict__ a, bool need_max, bool need_min, double& lmax, double& lmin,
                                                  double (&vmax)[D], double (&vmin)[D]) {
    constexpr int T = tri_size(D);
    double m[T];
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) { constexpr int c = decltype(cc)::value; m[tri(r, c)] = 0.5 * (a[r * D + c] + a[c * D + r]); });
    });
    double dg[D], e[D], ihh[D >= 3 ? D - 2 : 1], lam[D], sub[D], none[D];
    tridiagonalize_reflectors<D>(m, dg, e, ihh);
    static_for<D>([&](auto kk) {
        lam[decltype(kk)::value] = dg[decltype(kk)::value];
        sub[decltype(kk)::value] = e[decltype(kk)::value];
    });
    tridiag_ql_vectors<D, 1, false>(lam, sub, none);
    lmax = lam[0];
    lmin = lam[0];
    static_for<D - 1>([&](auto kk) {
        constexpr int c = decltype(kk)::value + 1;
        lmax = lam[c] > lmax ? lam[c] : lmax;
        lmin = lam[c] < lmin ? lam[c] : lmin;
    });
    auto back = [&](double (&z)[D]) {          // v = H_0 ... H_{D-3} z: the reflectors from the last to the first (as tridiagonalize_q builds Q)
        static_for_down<D - 3, 0>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            constexpr int n = D - k - 1;
            double t = 0.0;
            static_for<n>([&](auto rr) { constexpr int r = k + 1 + decltype(rr)::value; t = __builtin_fma(m[tri(r, k)], z[r], t); });
            t *= ihh[k];
            static_for<n>([&](auto rr) { constexpr int r = k + 1 + decltype(rr)::value; z[r] = __builtin_fma(-m[tri(r, k)], t, z[r]); });
        });
    };
    if (need_max) {
        tridiag_twisted_vector<D>(dg, e, lmax, vmax);
        back(vmax);
    }
    if (need_min) {
        tridiag_twisted_vector<D>(dg, e, lmin, vmin);
        back(vmin);
    }
}

// values and WHITENED Riemannian gradients of the built-in constraints at x (L = chol x already in the workspace):
// f = bound - lambda_max: egrad = -v v^T, rgrad = x egrad x, whitened L^-1 rgrad L^-T = -(L^T v)(L^T v)^T  (and + for lambda_min - bound);
// nested kinds: egrad = -+ G (G = d lambda / d x from nested_extremes_body), whitened = L^T egrad L.
template <int D>
__device__ __forceinline__ void builtin_constraints(const double* __restrict__ x, const TcgWs& w, int64_t i, int64_t R,
                                                    const BuiltinCons& B, double* nlds) {
    constexpr int dd = D * D;
    const double* L = w.chol + i * dd;
    if (builtin_has_kind(B, true)) {
        const NestedExtremesOut ne = nested_extremes_body<true, (D >= kWaveEighMinDim), kTrNestedMaxDim>(x, B.lift_w, B.lift_p, B.lift_x0, B.big_dim, D, nlds, true);
        for (int k = 0; k < B.n; ++k) {
            if (B.kind[k] < 2) continue;
            const bool want_max = B.kind[k] == 2;
            const double* G = ne.grad + (want_max ? 0 : dd);
            const double sign = want_max ? -1.0 : 1.0;
            if (threadIdx.x == 0) w.fc[i * B.n + k] = want_max ? B.bound[k] - ne.lam[0] : ne.lam[1] - B.bound[k];
            double* out = w.gc_w + ((int64_t)k * R + i) * dd;
            for (int e = threadIdx.x; e < dd; e += 64) {
                const int r = e / D, c = e - r * D;
                double s = 0.0;
                for (int a = 0; a < D; ++a) {
                    double t = 0.0;
                    for (int b = 0; b < D; ++b) t = __builtin_fma(G[a * D + b], L[b * D + c], t);
                    s = __builtin_fma(L[a * D + r], t, s);
                }
                out[e] = sign * s;
            }
        }
        __syncthreads();
    }
    if (!builtin_has_kind(B, false)) return;
    bool need_max = false, need_min = false;
    for (int k = 0; k < B.n; ++k) {
        need_max = need_max || B.kind[k] == 0;
        need_min = need_min || B.kind[k] == 1;
    }
    double lmax, lmin, vmax[D], vmin[D];
#ifndef GABO_BUILTIN_TWISTED      /* the full decomposition with all D vectors (rounds 3-6).  -DGABO_BUILTIN_TWISTED: eig_extreme_pairs - eigenvalue-only QL +
                                     one twisted-factorisation vector per extreme; 15.6 k -> 13.6 k cycles in the instrumented build, 694 against 691 us for
                                     the whole solve in the product build (tools/ab_solve_kernels.sh): not worth another numerical route */
    {
        double lam[D], v[dd];
        eig_extremes<D>(x, lam, v);
        lmax = lam[0], lmin = lam[0];
        static_for<D>([&](auto rr) { vmax[decltype(rr)::value] = v[decltype(rr)::value * D]; vmin[decltype(rr)::value] = v[decltype(rr)::value * D]; });
        static_for<D - 1>([&](auto kk) {
            constexpr int c = decltype(kk)::value + 1;
            const bool up = lam[c] > lmax, dn = lam[c] < lmin;
            lmax = up ? lam[c] : lmax;
            lmin = dn ? lam[c] : lmin;
            static_for<D>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                vmax[r] = up ? v[r * D + c] : vmax[r];
                vmin[r] = dn ? v[r * D + c] : vmin[r];
            });
        });
    }
#else
    eig_extreme_pairs<D>(x, need_max, need_min, lmax, lmin, vmax, vmin);
#endif
    for (int k = 0; k < B.n; ++k) {
        if (B.kind[k] >= 2) continue;
        const bool want_max = B.kind[k] == 0;
        const double best = want_max ? lmax : lmin;
        double vec[D];
        static_for<D>([&](auto rr) { vec[decltype(rr)::value] = want_max ? vmax[decltype(rr)::value] : vmin[decltype(rr)::value]; });
        double u[D];       // L^T v
        static_for<D>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            double sacc = 0.0;
            static_for<D - c>([&](auto rr) { constexpr int r = c + decltype(rr)::value; sacc = __builtin_fma(L[r * D + c], vec[r], sacc); });
            u[c] = sacc;
        });
        if (threadIdx.x == 0) {
            const double sign = want_max ? -1.0 : 1.0;
            w.fc[i * B.n + k] = want_max ? B.bound[k] - best : best - B.bound[k];
            double* out = w.gc_w + ((int64_t)k * R + i) * dd;
            static_for<D>([&](auto rr) {
                static_for<D>([&](auto cc) { out[decltype(rr)::value * D + decltype(cc)::value] = sign * u[decltype(rr)::value] * u[decltype(cc)::value]; });
            });
        }
    }
}

// strict variant: does the proposal violate a built-in constraint?  (constrained_trust_regions.py:932-951)
template <int D>
__device__ __forceinline__ bool builtin_infeasible(const double* __restrict__ xp, const BuiltinCons& B, double* nlds) {
    bool bad = false;
    if (builtin_has_kind(B, true)) {
        const NestedExtremesOut ne = nested_extremes_body<true, (D >= kWaveEighMinDim), kTrNestedMaxDim>(xp, B.lift_w, B.lift_p, B.lift_x0, B.big_dim, D, nlds, false);
        const double nmax = ne.lam[0], nmin = ne.lam[1];
        for (int k = 0; k < B.n; ++k) {
            if (B.kind[k] < 2) continue;
            const double f = B.kind[k] == 2 ? B.bound[k] - nmax : nmin - B.bound[k];
            bad = bad || (f < 0.0);
        }
        __syncthreads();
    }
    if (!builtin_has_kind(B, false)) return bad;
    double lmax, lmin, vmax[D], vmin[D];
    eig_extreme_pairs<D>(xp, false, false, lmax, lmin, vmax, vmin);          // (the eigenvalues alone: a third of the decomposition's instructions)
    for (int k = 0; k < B.n; ++k) {
        if (B.kind[k] >= 2) continue;
        const double f = B.kind[k] == 0 ? B.bound[k] - lmax : lmin - B.bound[k];
        bad = bad || (f < 0.0);
    }
    return bad;
}

// First part of a trust-region iteration for restart i: the state of truncated CG at x (tcg_begin) and the built-in constraints there.
template <int D>
__device__ __forceinline__ void tr_begin_part(const double* __restrict__ x, const double* __restrict__ g, double delta, const double* __restrict__ gc,
                                              const double* __restrict__ fc, const TrWs& t, int64_t i, int64_t R, int C, double* mats,
                                              int* __restrict__ status, const BuiltinCons* builtin, bool x_unchanged, double* nlds) {
    const TcgWs& w = t.tcg;
    GABO_TICK(1);
#ifdef GABO_TCG_BEGIN_LDS      /* A/B: the LDS-phased form everywhere (rounds 2-5) */
    tcg_begin(x, g, gc, fc, true, delta, w, i, R, D, C, status, mats, x_unchanged);
#else
    if constexpr (D <= 8) {
        if (x_unchanged) tcg_begin(x, g, gc, fc, true, delta, w, i, R, D, C, status, mats, true);
        else tcg_begin_reg<D>(x, g, gc, fc, true, delta, w, i, R, C, status, mats);
    } else {
        tcg_begin(x, g, gc, fc, true, delta, w, i, R, D, C, status, mats, x_unchanged);
    }
#endif
    __syncthreads();
    GABO_TICK(2);
    if constexpr (D <= 8) {
        // (x_unchanged: the previous proposal of this launch was rejected, so the constraint values and whitened gradients in the
        // workspace are still those of x: skip the eigen-solve)
        if (builtin != nullptr && builtin->n > 0 && !x_unchanged) {
            builtin_constraints<D>(x, w, i, R, *builtin, nlds);
            __syncthreads();
        }
    }
}

// Last part before the evaluation at the proposal: x+ = L expm(eta~) L^T, its Mandel vector, the model decrease.  Returns true when the step is the
// previous one again (see below): nothing was rebuilt, the previous proposal and its value stand.
template <int D>
__device__ __forceinline__ bool tr_build_proposal(const TrWs& t, double* __restrict__ x_prop, int64_t i, bool x_unchanged, double* step_cache,
                                                  double* mats) {
    constexpr int T = tri_size(D);
    constexpr int dd = D * D;
    const TcgWs& w = t.tcg;
    // ---- proposal x+ = L expm(eta~) L^T and the model decrease -<g, eta> - 1/2 <eta, H eta> (whitened Frobenius dots)
    double* M0 = mats;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* cs = mats + 5 * dd;
    const double* etaw = w.eta_w + i * dd;
    const double ge = wave_dot(w.g_w + i * dd, etaw, dd);
    const double ehe = wave_dot(etaw, w.heta_w + i * dd, dd);
    if (threadIdx.x == 0) t.rhoden[i] = -ge - 0.5 * ehe;
    // The same step again.  A restart that sits on a constraint bound can have its tCG step set by the LINEARISED CONSTRAINT (stop reason
    // "reached constraints"), whatever the trust radius: the rejected proposal quarters the radius, tCG runs again from the same x, gradient and
    // constraints and returns the SAME eta, bit for bit - for every one of the remaining iterations (config 4: 4 of 512 restarts do this 98 times
    // in a row, tools/tr_eta_probe.py; they were the whole duration of the launch).  With x unchanged and eta~ identical the proposal, its
    // acquisition value and therefore the rho test are those of the previous iteration, which are still in the workspace: skip the matrix
    // exponential, the congruence and the evaluation.  `step_cache`: T + 1 doubles of LDS (the last eta~ as a packed triangle, a valid flag).
    if (step_cache != nullptr) {
        bool same = x_unchanged && step_cache[T] != 0.0;
        for (int e = threadIdx.x; e < dd; e += 64) {
            const int r = e / D, c = e - r * D;
            const double sym = 0.5 * (etaw[r * D + c] + etaw[c * D + r]);
            if (r >= c) {
                same = same && (sym == step_cache[tri(r, c)]);
            }
        }
        same = __builtin_amdgcn_ballot_w64(!same) == 0;       // every lane agrees (lanes beyond D^2 vote with the flag alone)
        __syncthreads();                                        // all compares done before the cache is rewritten
        for (int e = threadIdx.x; e < dd; e += 64) {
            const int r = e / D, c = e - r * D;
            if (r >= c) step_cache[tri(r, c)] = 0.5 * (etaw[r * D + c] + etaw[c * D + r]);
        }
        if (threadIdx.x == 0) step_cache[T] = 1.0;
        if (same) {
            __syncthreads();
            return true;
        }
    }
    lds_load(w.chol + i * dd, M0, D);
    if constexpr (D <= 8) {
        // expm(eta~) from the register eigen-decomposition (every lane redundantly, no barriers); lane 0 publishes E
        double m[T], v[D * D];
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            static_for<r + 1>([&](auto cc) { constexpr int c = decltype(cc)::value; m[tri(r, c)] = 0.5 * (etaw[r * D + c] + etaw[c * D + r]); });
        });
        double lam_e[D];
        sym_eig_reg<D>(m, lam_e, v);
        double ex[D];
        static_for<D>([&](auto kk) { ex[decltype(kk)::value] = exp(lam_e[decltype(kk)::value]); });
        if (threadIdx.x == 0) {
            static_for<D>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                static_for<r + 1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    double f = 0.0;
                    static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; f = __builtin_fma(v[r * D + k] * ex[k], v[c * D + k], f); });
                    M3[r * D + c] = f;
                    M3[c * D + r] = f;
                });
            });
        }
        __syncthreads();
    } else {
        lds_load(etaw, M1, D);
        lds_jacobi(M1, M2, cs, D);
        lds_fun_from_eig(M1, M2, M3, D, FN_EXP, cs);
    }
    lds_congruence(M0, M3, M1, M2, D);
    lds_symmetrize(M1, M2, D);
    lds_store(M1, x_prop, D);
    double* xpm = t.xp_mandel + i * T;
    for (int e = threadIdx.x; e < T; e += 64) {
        int k = 0;
        while (k + 1 < D && (k + 1) * D - (k + 1) * k / 2 <= e) ++k;
        int cc = e - (k * D - k * (k - 1) / 2);
        int r = cc + k;
        xpm[e] = (k == 0) ? M1[r * D + cc] : kSqrt2 * M1[r * D + cc];
    }
    __syncthreads();
    return false;
}

// tCG + proposal + acquisition at the proposal for restart i (one wave).  gc/fc: host-evaluated constraints, or null with
// `builtin` set (then the wave evaluates them itself).  mats: 5 D^2 + kJacobiScratch doubles of LDS, dyn: 3 n doubles.
// Returns the number of tCG iterations it ran.  x_unchanged: the previous call for this restart (same launch) ended in a rejected proposal;
// fd0_kept: ... and ran exactly ONE tCG iteration, so the first FD point's E = expm(c delta~) and c are still in the workspace.
template <int D, int METRIC>
__device__ __forceinline__ int tr_propose_body(const double* __restrict__ x, const double* __restrict__ g, double delta,
                                                const double* __restrict__ gc, const double* __restrict__ fc, const AcqParams& P,
                                                const TrWs& t, double* __restrict__ x_prop, int64_t i, int64_t R, int C, int neq,
                                                double delta_cons, double theta, double kappa, int mininner, int maxinner,
                                                AcqLds<D>& acq, double* mats, double* dyn, int* __restrict__ status,
                                                const BuiltinCons* builtin, bool x_unchanged = false, bool fd0_kept = false,
                                                double* nlds = nullptr, bool value_only = false, double* step_cache = nullptr) {
    constexpr int T = tri_size(D);
    const TcgWs& w = t.tcg;
    double* xfd = t.x_fd + i * T;
    double* egfd = t.eg_fd + i * T;
    double* F = t.F + i * T * P.n;
    tr_begin_part<D>(x, g, delta, gc, fc, t, i, R, C, mats, status, builtin, x_unchanged, nlds);
    double* egfd0 = t.eg_fd0 + i * T;
    GABO_TICK(3);
    int inner = 0;
    for (int it = 0; it < maxinner; ++it) {
        ++inner;
        // (same x, g, preconditioner => same first direction, same FD point: when nothing has overwritten E and c since, skip it - the
        // acquisition gradient there is kept too, below)
        if (!(it == 0 && x_unchanged && fd0_kept)) tcg_fd_point(w, i, D, xfd, mats);
        __syncthreads();
        GABO_TICK(4);
        // tCG restarts from eta = 0 with the same x, g and preconditioner after a rejected proposal (only the radius changed), so its
        // first direction, first FD point and the acquisition gradient there are bit for bit those of the previous iteration: keep
        // that gradient instead of evaluating the acquisition again (a restart that sits on a bound does one tCG step per
        // iteration - this is half of its acquisition evaluations)
        double* eg_it = (it == 0) ? egfd0 : egfd;
        if (!(it == 0 && x_unchanged)) acq_eval_any<D, METRIC>(xfd, P, t.val_fd + i, eg_it, F, acq, dyn, status, i + w.index_base);
        __syncthreads();
        GABO_TICK(5);
        const bool running = tcg_step(w, i, R, D, C, eg_it, neq, delta_cons, theta, kappa, mininner, it, mats);
        __syncthreads();
        GABO_TICK(6);
        if (!running) break;
    }
    if (tr_build_proposal<D>(t, x_prop, i, x_unchanged, step_cache, mats)) {
        GABO_TICK(7);
        GABO_TICK(8);
        return inner;
    }
    double* xpm = t.xp_mandel + i * T;
    GABO_TICK(7);
    // value_only: the acquisition VALUE at the proposal now, its gradient only if the proposal is accepted (tr_solve: a restart that has just
    // had a proposal rejected will most likely have the next one rejected too, and a rejected proposal's gradient is never looked at)
    acq_eval_any<D, METRIC>(xpm, P, t.fx_prop + i, value_only ? nullptr : t.eg_prop + i * T, F, acq, dyn, status, i + w.index_base);
    GABO_TICK(8);
    return inner;
}

// The exact-GP factors L^-1 and L^-T (n^2 doubles each) are read by every acquisition evaluation of the launch - two triangular
// matrix-vector products whose per-lane loops wait for one L2 round trip per term (clock instrumentation, tools/tr_clocks.py: 15 k
// + 16 k of the 55-64 k cycles of an evaluation at n = 50).  When they fit, the wave copies them into LDS once per launch and the
// evaluations read them from there.  Dynamic LDS: 3 n doubles of scratch, then 2 n^2 doubles when `stage_gp`.
static __device__ __forceinline__ AcqParams stage_gp_factors(const AcqParams& P, double* dyn, int stage_gp) {
    AcqParams Ps = P;
    if (stage_gp && P.linv && P.linv_t) {
        const int64_t nn = P.n * P.n;
        double* gl = dyn + 3 * P.n;
        const bool one = P.linv == P.linv_t;          // the symmetric inverse handed over for both (spd_acq_body.hpp): staged once
        for (int64_t e = threadIdx.x; e < nn; e += blockDim.x) {
            gl[e] = P.linv[e];
            if (!one) gl[nn + e] = P.linv_t[e];
        }
        __syncthreads();
        Ps.linv = gl;
        Ps.linv_t = one ? gl : gl + nn;
    }
    return Ps;
}

// dynamic LDS bytes of the trust-region kernels and whether the GP factors are staged: when they fit (total LDS of a block below 64 KB)
// and the launch is in the latency regime.  With thousands of restarts the CUs are full and LDS capacity limits the blocks per CU:
// measured per propose launch, staged vs not: 113 vs 137 us at 512 restarts, 242 vs 209 at 2048, 757 vs 689 at 8192.
// factors: how many n x n matrices are staged - 2 (L^-1 and L^-T), or 1 when the caller handed over the symmetric inverse for both (tr_factor_count).  With one
// (20 KB at n = 50) the block's LDS no longer decides how many blocks a CU holds (four waves of 512 registers do), so the restart limit does not apply:
// 1024 restarts run in one round instead of two (sweep 1.60 -> 0.95 ms), and the LDS-resident forms serve every restart count.
static __host__ __device__ inline int tr_factor_count(const AcqParams& P) { return (P.linv != nullptr && P.linv == P.linv_t) ? 1 : 2; }
static inline size_t tr_dynamic_lds(int64_t n, int64_t restarts, int* stage_gp, int factors = 2) {
    const size_t base = (size_t)(3 * n) * sizeof(double), staged = (size_t)(factors * n * n) * sizeof(double);
    *stage_gp = (base + staged <= 48 * 1024 && (restarts <= 1024 || (factors == 1 && base + staged <= 24 * 1024))) ? 1 : 0;
    return base + (*stage_gp ? staged : 0);
}

// The single-launch solve owns its restart from the first iteration to the last: nothing in the workspace has to survive the launch, so
// the whole per-restart workspace (whitened tCG vectors, FD point, proposal, the logm spill: ~1 k doubles at d = 5, n = 50) can live in
// the block's LDS instead of L2 - every phase of an iteration starts with dependent loads of that state.  Only in the latency regime and
// when it fits next to the staged GP factors (static LDS of the kernel is below 12 KB for d <= 8).
// nested_bytes: LDS of the nested eigenvalue constraints (nested_extremes_lds_doubles), placed last; *nested_off receives its offset.
static inline size_t tr_solve_dynamic_lds(int64_t n, int64_t restarts, int d, int C, int* stage_gp, int* ws_lds, size_t nested_bytes = 0,
                                          int* nested_off = nullptr, int factors = 2) {
    size_t bytes = tr_dynamic_lds(n, restarts, stage_gp, factors);
    bytes = (bytes + 15) & ~(size_t)15;
    const size_t ws = tr_layout(nullptr, 1, d, C, n).bytes + 16;
    // (beyond 1024 restarts only while four blocks still fit a CU's 160 KB next to the ~10 KB of static LDS: 30 KB of dynamic LDS)
    *ws_lds = ((restarts <= 1024 || (factors == 1 && bytes + ws + nested_bytes <= 30 * 1024)) && bytes + ws + nested_bytes <= 52 * 1024) ? 1 : 0;
    bytes += (*ws_lds ? ws : 0);
    bytes = (bytes + 15) & ~(size_t)15;
    if (nested_off) *nested_off = (int)bytes;
    return bytes + nested_bytes;
}

template <int D, int METRIC>
__global__ __launch_bounds__(64) void spd_tr_propose_kernel(const double* __restrict__ x, const double* __restrict__ g,
                                                            const double* __restrict__ delta_tr, const uint8_t* __restrict__ active,
                                                            const double* __restrict__ gc, const double* __restrict__ fc,
                                                            AcqParams P, void* wsbase, double* __restrict__ x_prop, int64_t R, int C,
                                                            int neq, double delta_cons, double theta, double kappa, int mininner,
                                                            int maxinner, int* __restrict__ any_active, int* __restrict__ status,
                                                            int stage_gp) {
    constexpr int dd = D * D;
    __shared__ AcqLds<D> acq;
    __shared__ __attribute__((aligned(16))) double mats[5 * dd + kJacobiScratch];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    if (i == 0 && threadIdx.x == 0) *any_active = 0;          // set again by the update kernel
    if (active[i] == 0) return;
    TrWs t = tr_layout(wsbase, R, D, C, P.n);
    const AcqParams Ps = stage_gp_factors(P, dyn, stage_gp);
    tr_propose_body<D, METRIC>(x + i * dd, g + i * dd, delta_tr[i], gc, fc, Ps, t, x_prop + i * dd, i, R, C, neq, delta_cons, theta, kappa,
                               mininner, maxinner, acq, mats, dyn, status, nullptr);
}

// the acceptance test of the update below on its own (robust_trust_regions.py:236-300): model decrease and rho > rho_prime, with the
// regularisation of rho (robust_trust_regions.py:262-271).  The same statements as in tr_update_body, so that the two agree bit for bit.
static __device__ __forceinline__ bool tr_would_accept(double fx0, double fx_prop, double rhoden_raw, bool inval, double rho_prime,
                                                       double rho_regularization) {
    const double fxp = inval ? __builtin_inf() : fx_prop;
    const double rho_reg = (__builtin_fabs(fx0) > 1.0 ? __builtin_fabs(fx0) : 1.0) * 2.220446049250313e-16 * rho_regularization;
    const double rhonum = (fx0 - fxp) + rho_reg;
    const double rhoden = rhoden_raw + rho_reg;
    const bool model_decreased = rhoden >= 0.0;
    const double rho = rhoden == 0.0 ? __builtin_nan("") : rhonum / rhoden;
    return model_decreased && rho > rho_prime;
}

// rho test and state update of restart i (robust_trust_regions.py:236-330; same algebra as BatchedTrustRegions.solve).
// lds: 4 d^2 doubles.  Returns true while the restart stays active.
static __device__ bool tr_update_body(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g, double* __restrict__ ng,
                                      double* __restrict__ delta_tr, int64_t* __restrict__ iters, bool inval,
                                      const double* __restrict__ x_prop, const TrWs& t, int64_t i, int d, int C, double delta_bar,
                                      double rho_prime, double rho_regularization, double mingradnorm, int64_t maxiter, double* lds,
                                      bool* accepted = nullptr) {
    const int dd = d * d;
    double* M0 = lds;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    const double fx0 = *fx;
    const double fxp = inval ? __builtin_inf() : t.fx_prop[i];
    const double rho_reg = (__builtin_fabs(fx0) > 1.0 ? __builtin_fabs(fx0) : 1.0) * 2.220446049250313e-16 * rho_regularization;
    const double rhonum = (fx0 - fxp) + rho_reg;
    const double rhoden = t.rhoden[i] + rho_reg;
    const bool model_decreased = rhoden >= 0.0;
    const double rho = rhoden == 0.0 ? __builtin_nan("") : rhonum / rhoden;
    const bool shrink = (rho < 0.25) || !model_decreased || (rho != rho) || inval;
    const int stop_inner = t.tcg.stop[i];
    const bool boundary = stop_inner == TCG_NEGATIVE_CURVATURE || stop_inner == TCG_EXCEEDED_TR ||
                          (C > 0 && stop_inner == TCG_REACHED_CONSTRAINTS);
    const bool grow = !shrink && rho > 0.75 && boundary;
    const double D0 = *delta_tr;
    const double Dn = shrink ? D0 / 4 : (grow ? (2 * D0 < delta_bar ? 2 * D0 : delta_bar) : D0);
    const bool accept = model_decreased && rho > rho_prime;
    if (accepted) *accepted = accept;                  // (the same value in every lane)
    double ngi = *ng;
    const int64_t it = *iters + 1;
    __syncthreads();                 // every lane has read the scalars before lane 0 rewrites them
    if (accept) {
        // x <- x+, g <- x+ sym(egrad) x+ (egrad2rgrad), ||g||_x = sqrt(tr(S x S x))
        lds_load(x_prop, M0, d);
        lds_from_mandel(t.eg_prop + i * (int64_t)(d * (d + 1) / 2), M1, d);
        lds_mm(M1, M0, M2, d, false, false);          // P = S X
        lds_mm(M0, M2, M3, d, false, false);          // X S X
        double s = 0.0;
        for (int e = threadIdx.x; e < dd; e += 64) {
            int r = e / d, c = e - r * d;
            s = __builtin_fma(M2[e], M2[c * d + r], s);
            x[e] = M0[e];
            g[e] = 0.5 * (M3[e] + M3[c * d + r]);
        }
        s = wave_sum(s);
        ngi = __builtin_sqrt(s > 0.0 ? s : 0.0);
    }
    if (threadIdx.x == 0) {
        *delta_tr = Dn;
        if (accept) { *fx = fxp; *ng = ngi; }
        *iters = it;
    }
    __syncthreads();
    return !(ngi < mingradnorm || it >= maxiter);
}

// The first iteration of tcg_step_core as a function of the radius alone: which way tCG leaves (or -1: it takes the tentative step and goes on) and the step
// along delta_0, from the quantities that do not change while x stands (the scalars of tcg_begin, <delta_0, H delta_0>, the constraints' values and their
// directional derivatives).  The statements of tcg_step_core with eta = 0.
struct TrFirstStep { int stop; double step; };
static __device__ __forceinline__ TrFirstStep tr_first_step(double Delta, const double* sc, double d_Hd, const double* fcl, const double* fpe, const double* fpd,
                                                            int C, double delta_cons) {
    const double e_Pe = sc[SC_E_PE], e_Pd = sc[SC_E_PD], d_Pd = sc[SC_D_PD], z_r = sc[SC_Z_R];
    const double dc2 = delta_cons * delta_cons;
    const bool nz = d_Hd != 0.0;
    const double alpha = nz ? z_r / d_Hd : 0.0;
    const double e_Pe_new = nz ? e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd : e_Pe;
    const double Delta2 = Delta * Delta;
    TrFirstStep r{-1, 0.0};
    if (d_Hd <= 0.0 || e_Pe_new >= Delta2) {
        double tau = (-e_Pd + __builtin_sqrt(e_Pd * e_Pd + d_Pd * (Delta2 - e_Pe))) / d_Pd;
        r.stop = d_Hd <= 0.0 ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;
        if (C > 0) {
            if (tau != tau) tau = 0.0;
            ConsStep cst = cons_step(tau, fcl, fpe, fpd, C, 0, dc2);
            if (cst.cin > dc2) {
                tau = cst.tau;
                if (d_Hd > 0.0) r.stop = TCG_REACHED_CONSTRAINTS;
            }
        }
        r.step = tau;
    } else if (C > 0) {
        ConsStep cst = cons_step(alpha, fcl, fpe, fpd, C, 0, dc2);
        if (cst.cin > dc2) { r.stop = TCG_REACHED_CONSTRAINTS; r.step = cst.tau; }
    }
    return r;
}

// Rejected again and again (single-launch solves, after an update that rejected a proposal whose tCG ran ONE iteration and stopped).  A rejected proposal
// quarters the radius and changes nothing else: the next iteration's tCG runs from the same x, gradient, delta_0, H delta_0 and constraints, and when it leaves
// in its first step with the SAME step along delta_0 - a restart that sits outside an eigenvalue bound has its step set by the linearised constraint whatever the
// radius: config 4 has such restarts reject 98 proposals in a row, 17 k cycles each, and they were the duration of the launch - eta, the proposal, its value,
// the model decrease and therefore the verdict are those of this iteration again.  What such an iteration does to the state is known - one more count, the
// radius quartered again (tr_update_body) - and is applied here for as many iterations as the first-step logic (scalars only) returns the same step: bit
// for bit the state those iterations would have left (tests/test_gpu_native_sweep.py compares with GABO_TR_NO_SHORTCUTS).  Hd: H delta_0~ as tcg_step left it
// (M4 of its LDS tile); the workspace still holds delta_0~, the scalars of tcg_begin and the constraints (tCG's stop path does not touch them).
// Returns the number of iterations applied; *still becomes false when maxiter is reached.
#ifdef GABO_TR_FF_INLINE
#define GABO_FF_INLINE __forceinline__
#else
#define GABO_FF_INLINE
#endif
static __device__ GABO_FF_INLINE int tr_repeat_rejected(const TcgWs& w, int64_t iw, int64_t Rw, int d, int C, const double* Hd, double delta_cons, double* __restrict__ delta_tr,
                                         int64_t* __restrict__ iters, int64_t maxiter, bool* still) {
    const int dd = d * d;
    const double* dl0 = w.delta_w + iw * dd;
    const double* sc = w.scal + iw * SC_COUNT;
    const double d_Hd = wave_dot(dl0, Hd, dd);
    double fcl[kMaxCons], fpe[kMaxCons], fpd[kMaxCons];
    for (int k = 0; k < C; ++k) {
        fcl[k] = w.fc[iw * C + k];
        fpe[k] = w.fcg_pe[iw * C + k];
        fpd[k] = wave_dot(w.gc_w + ((int64_t)k * Rw + iw) * dd, dl0, dd);
    }
    const TrFirstStep ref = tr_first_step(sc[SC_DELTA], sc, d_Hd, fcl, fpe, fpd, C, delta_cons);
    double Dk = *delta_tr;
    int64_t itk = *iters;
    int skipped = 0;
    while (ref.stop >= 0) {
        const TrFirstStep nxt = tr_first_step(Dk, sc, d_Hd, fcl, fpe, fpd, C, delta_cons);
        if (nxt.stop < 0 || !(nxt.step == ref.step)) break;
        ++itk;
        Dk = Dk / 4;
        ++skipped;
        if (itk >= maxiter) { *still = false; break; }
    }
    __syncthreads();                 // every lane has read the scalars before lane 0 rewrites them
    if (skipped > 0 && threadIdx.x == 0) {
        *delta_tr = Dk;
        *iters = itk;
    }
    __syncthreads();
    return skipped;
}

// The sweep's own start and end of a restart (gabo_spd_sweep_solve_rows, spd_sweep.hip): restart i begins at raw sample picked[i] of the scored table and
// the device itself does what gen_candidates_manifold does around the solver (manifold_optimize.py:170-205) - pre_processing_manifold (Mandel -> matrix),
// post_processing_manifold inside the cost (matrix -> Mandel), cost and Euclidean gradient at the start, [3P] egrad2rgrad and norm of pymanopt's
// PositiveDefinite (robust_trust_regions.py:148-158), radius and counters (spd_tr_start_kernel) - and, after the last iteration, the Mandel vector of the
// final iterate (tr_finish_body, inside the solve launch).  Rounds 4-5 issued these as ten launches in front of the solve and three behind it; the
// statements below are THOSE kernels' statements in the same order (mandel.hip, spd_acq_kernel, spd_manifold.hip OP_EGRAD2RGRAD / OP_NORM), so the bits
// are the ones the separate launches give.
struct TrStart {
    const double* raw_rows;     // null: the caller filled x, fx, g, ng, delta_tr, active, iters (gabo_spd_tr_solve)
    int64_t raw_stride;         // doubles per row of the table: [value, Mandel vector ...]
    const int64_t* picked;      // row index of every restart (device-visible: device memory or mapped host memory)
    double delta0;
    double* res_rows;           // restarts x (2 + T): final cost, iterations, final iterate as a Mandel vector
    double* res_host;           // the same rows in mapped host memory, or null
    int* status_host;           // mapped host copy of an error this launch reports, or null
    const int* skip_flag;       // device int, or null: non-zero = the selection in front of this launch picked nothing (it raised its fall-back flag):
                                // the start marks every restart inactive and the solve returns at once
};

template <int D, int METRIC>
__device__ __forceinline__ void tr_start_body(const TrStart& S, double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                              double* __restrict__ ng, const AcqParams& P, const TrWs& t, int64_t iw, AcqLds<D>& acq,
                                              double* mats, double* dyn, int* __restrict__ status) {
    constexpr int T = tri_size(D);
    constexpr int dd = D * D;
    double* M0 = mats;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    double* M4 = M3 + dd;
    const double* row = S.raw_rows + S.picked[blockIdx.x] * S.raw_stride + 1;
    for (int e = threadIdx.x; e < dd; e += 64) {          // mandel_to_matrix_kernel
        const int r = e / D, c = e - r * D;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        const double v = row[mandel_pos(D, hi, lo)];
        const double m = (r == c) ? v : v / kSqrt2;
        M0[e] = m;
        x[e] = m;
    }
    __syncthreads();
    double* xm = t.xp_mandel + iw * T;
    for (int e = threadIdx.x; e < T; e += 64) {           // matrix_to_mandel_kernel
        int k = 0;
        while (k + 1 < D && (k + 1) * D - (k + 1) * k / 2 <= e) ++k;
        const int c = e - (k * D - k * (k - 1) / 2);
        const int r = c + k;
        xm[e] = (k == 0) ? M0[r * D + c] : 0.5 * (kSqrt2 * M0[c * D + r] + kSqrt2 * M0[r * D + c]);
    }
    __syncthreads();
    double* egm = t.eg_prop + iw * T;
    acq_eval_any<D, METRIC>(xm, P, fx, egm, t.F + iw * T * P.n, acq, dyn, status, iw + t.tcg.index_base);
    __syncthreads();
    for (int e = threadIdx.x; e < dd; e += 64) {          // mandel_to_matrix_kernel on the gradient
        const int r = e / D, c = e - r * D;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        const double v = egm[mandel_pos(D, hi, lo)];
        M1[e] = (r == c) ? v : v / kSqrt2;
    }
    __syncthreads();
    lds_symmetrize(M1, M2, D);                             // OP_EGRAD2RGRAD: X sym(G) X
    lds_congruence(M0, M1, M3, M4, D);
    lds_store(M3, g, D);
    for (int e = threadIdx.x; e < dd; e += 64) M1[e] = M3[e];
    __syncthreads();
    lds_cholesky(M0, D);                                   // OP_NORM: ||g||_x^2 = ||L^-1 g L^-T||_F^2
    lds_tri_inverse(M0, M2, D);
    lds_congruence(M2, M1, M3, M4, D);
    if (threadIdx.x == 0) {
        double sacc = 0.0;
        for (int k = 0; k < dd; ++k) sacc = __builtin_fma(M3[k], M3[k], sacc);
        *ng = __builtin_sqrt(sacc > 0.0 ? sacc : 0.0);
    }
    __syncthreads();
}

template <int D>
__device__ __forceinline__ void tr_finish_body(const TrStart& S, const double* __restrict__ x, double fx, int64_t iters) {
    constexpr int T = tri_size(D);
    double* row = S.res_rows + (int64_t)blockIdx.x * (2 + T);
    double* hrow = S.res_host ? S.res_host + (int64_t)blockIdx.x * (2 + T) : nullptr;
    for (int e = threadIdx.x; e < T; e += 64) {           // matrix_to_mandel_kernel
        int k = 0;
        while (k + 1 < D && (k + 1) * D - (k + 1) * k / 2 <= e) ++k;
        const int c = e - (k * D - k * (k - 1) / 2);
        const int r = c + k;
        const double v = (k == 0) ? x[r * D + c] : 0.5 * (kSqrt2 * x[c * D + r] + kSqrt2 * x[r * D + c]);
        row[2 + e] = v;
        if (hrow) hrow[2 + e] = v;
    }
    if (threadIdx.x == 0) {
        row[0] = fx;
        row[1] = (double)iters;
        if (hrow) { hrow[0] = fx; hrow[1] = (double)iters; }
    }
}

// The whole trust-region solve of restart i in one launch: no host involvement between iterations.  Possible when the
// constraints are the built-in eigenvalue bounds (or there are none); D <= 8.
// LAT (the latency regime: <= 1024 restarts, everything fits): the GP factors AND the workspace are in LDS, known at compile time.  With
// the two as runtime flags every workspace / factor pointer is "LDS or global", i.e. a generic pointer, and its accesses are flat_load /
// flat_store - which count in vmcnt and lgkmcnt at once, so each dependent access is a full `s_waitcnt vmcnt(0) lgkmcnt(0)` drain through
// the memory pipeline (d = 5: 159 flat loads, 89 flat stores, 89 such drains in the kernel).  Specialised, they are ds_read / ds_write.
#ifndef GABO_TR_SOLVE_MIN_WAVES
#define GABO_TR_SOLVE_MIN_WAVES 1      /* A/B: 2 = at most 256 registers, two waves per SIMD (tools/ab_build.py) */
#endif
template <int D, int METRIC, bool LAT = false>
__global__ __launch_bounds__(64, GABO_TR_SOLVE_MIN_WAVES) void spd_tr_solve_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                          double* __restrict__ ng, double* __restrict__ delta_tr,
                                                          uint8_t* __restrict__ active, int64_t* __restrict__ iters, AcqParams P,
                                                          BuiltinCons B, void* wsbase, int64_t R, double delta_cons, double theta,
                                                          double kappa, int mininner, int maxinner, double delta_bar, double rho_prime,
                                                          double rho_regularization, double mingradnorm, int64_t maxiter,
                                                          int* __restrict__ status, int stage_gp, int ws_lds, int nested_off, int shortcuts,
                                                          double* __restrict__ rec, int64_t rec_cap, TrStart S) {
    static_assert(D <= 8, "built-in constraints use the register eigen-solver");
    constexpr int dd = D * D;
    __shared__ AcqLds<D> acq;
    // (five tiles; the eigen-solvers' LDS scratch behind them is only used above d = 8.  Every byte counts here: with the symmetric inverse staged a block is
    // 39 KB at d = 5, n = 50, and FOUR blocks - one wave of 512 registers per SIMD - have to fit a CU's 160 KB for 1024 restarts to run in one round)
    __shared__ __attribute__((aligned(16))) double mats[5 * dd + 8];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    // (the sweep's START is a launch of its own, spd_tr_start_kernel below: compiled into this kernel - one more inlined acquisition evaluation in a
    // function that already fills 512 registers and 712 bytes of scratch per lane - it made EVERY solve 8 % slower, run or not: 756 against 699 us
    // at 64 restarts, tools/ab_solve_kernels.sh.  The END - the result row - is a dozen stores and stays here.)
    const bool own_finish = S.res_rows != nullptr;            // (uniform over the launch)
    if (active[i] == 0) {
        if (own_finish) tr_finish_body<D>(S, x + i * dd, fx[i], iters[i]);
        return;
    }
    const int C = B.n;
    if constexpr (LAT) {
        stage_gp = 1;
        ws_lds = 1;
    }
    AcqParams Ps = P;
    if constexpr (LAT) {          // (the dispatcher has checked that the factors exist)
        const int64_t nn = P.n * P.n;
        double* gl = dyn + 3 * P.n;
        const bool one = P.linv == P.linv_t;          // the symmetric inverse handed over for both (spd_acq_body.hpp): staged once
        for (int64_t e = threadIdx.x; e < nn; e += blockDim.x) {
            gl[e] = P.linv[e];
            if (!one) gl[nn + e] = P.linv_t[e];
        }
        __syncthreads();
        Ps.linv = gl;
        Ps.linv_t = one ? gl : gl + nn;
    } else {
        Ps = stage_gp_factors(P, dyn, stage_gp);
    }
    // workspace: this block's slice of the caller's buffer, or (ws_lds) a private copy of the layout for ONE restart in LDS
    TrWs t;
    int64_t iw = i, Rw = R;
    if (ws_lds) {
        size_t off = (size_t)(3 * P.n + (stage_gp ? tr_factor_count(P) * P.n * P.n : 0)) * sizeof(double);
        off = (off + 15) & ~(size_t)15;
        char* base = (char*)dyn + off;
        t = tr_layout(base, 1, D, C, P.n);
        for (size_t e = threadIdx.x; e < t.bytes / sizeof(double); e += blockDim.x) ((double*)base)[e] = 0.0;
        t.tcg.index_base = i;
        iw = 0;
        Rw = 1;
        __syncthreads();
    } else {
        t = tr_layout(wsbase, R, D, C, P.n);
    }
    double* xp = t.xp_mat + iw * dd;
    double* nlds = reinterpret_cast<double*>(reinterpret_cast<char*>(dyn) + nested_off);      // (used by nested constraint kinds only)
    bool cons_fresh = false;          // wave-uniform: the constraints in the workspace belong to the current x
    int last_inner = 0;               // tCG iterations of the previous trust-region iteration
    constexpr int T_ = tri_size(D);
#ifndef GABO_TR_NO_STEP_CACHE
    __shared__ double step_cache_store[T_ + 1];     // the last proposal's eta~ and a valid flag: see "The same step again" in tr_propose_body
    if (threadIdx.x == 0) step_cache_store[T_] = 0.0;
    __syncthreads();
    double* const step_cache = shortcuts != 0 ? step_cache_store : nullptr;
#else
    double* const step_cache = nullptr;
#endif
    int64_t rec_k = rec != nullptr ? iters[i] : 0;      // gabo_tr_solve_record: index of the outer iteration being recorded
    for (;;) {
        // Value first after a rejection.  The restarts that set this launch's duration are the ones whose proposals are rejected again and again
        // (config 4: 4 of 512 restarts sit on the eigenvalue bound and have 99 of their 100 proposals rejected, the radius ending at 2.4e-60, while
        // the other 508 finish within 12 iterations - tools/tr_accept_stats.py): their iteration is tCG (cached begin), proposal, acquisition at the
        // proposal, update - and the acquisition GRADIENT there, a third of that evaluation (eigenvectors, logm of every pair, the second
        // triangular product, the adjoint chain), is never used.  So the iteration after a rejected one evaluates the value alone (with the
        // eigenvalue recurrence of the full evaluation: the same bits, sym_eig_reg_values) and the gradient only if the acceptance test passes.
#ifdef GABO_TR_NO_LAZY_GRADIENT      /* A/B: value and gradient together in every iteration (rounds 1-4) */
        const bool lazy = false;
#else
        const bool lazy = cons_fresh && shortcuts != 0;
#endif
#ifndef GABO_TR_SINGLE_SITE     /* the iteration as tr_propose_body (two inlined acquisition evaluations) + a third evaluation here */
        last_inner = tr_propose_body<D, METRIC>(x + i * dd, g + i * dd, delta_tr[i], nullptr, nullptr, Ps, t, xp, iw, Rw, C, 0, delta_cons, theta,
                                                kappa, mininner, maxinner, acq, mats, dyn, status, &B, cons_fresh, last_inner == 1, nlds, lazy, step_cache);
        __syncthreads();
        if (rec != nullptr && rec_k < rec_cap) {          // (the iterate, its radius and the stop reason of the tCG run that made the proposal)
            double* rr = rec + (rec_k * R + i) * (dd + 2);
            for (int e = threadIdx.x; e < dd; e += 64) rr[e] = x[i * dd + e];
            if (threadIdx.x == 0) {
                rr[dd] = delta_tr[i];
                rr[dd + 1] = (double)t.tcg.stop[iw];
            }
        }
        ++rec_k;
        const bool inval = (B.strict && C > 0) ? builtin_infeasible<D>(xp, B, nlds) : false;
        if (lazy && tr_would_accept(fx[i], t.fx_prop[iw], t.rhoden[iw], inval, rho_prime, rho_regularization)) {
            acq_eval_any<D, METRIC>(t.xp_mandel + iw * T_, Ps, t.fx_prop + iw, t.eg_prop + iw * T_, t.F + iw * T_ * Ps.n, acq, dyn, status,
                                    iw + t.tcg.index_base);
            __syncthreads();
        }
#else
        // An experiment of round 6 (-DGABO_TR_SINGLE_SITE; bit-identical results, 720 against 700 us for the solve at 64 restarts: not the default).
        // The same iteration with ONE call site of the acquisition evaluation: the evaluations of an iteration - at the FD point of every tCG step,
        // at the proposal, again at the proposal for its gradient once it is known to be accepted - are trips of one loop whose body is "prepare the
        // next evaluation, evaluate, consume".  acq_eval is inlined wherever it is called, and this kernel sits on a register cliff (512 registers,
        // 712 B of scratch per lane): every copy moves the whole kernel (a fourth one, the sweep's start, cost 8 %: see above).  Same statements in
        // the same order as tr_propose_body + the block above: same bits.
        bool inval = false;
        {
            const bool x_unchanged = cons_fresh, fd0_kept = last_inner == 1;
            const TcgWs& w = t.tcg;
            double* xfd = t.x_fd + iw * T_;
            double* egfd = t.eg_fd + iw * T_;
            double* egfd0 = t.eg_fd0 + iw * T_;
            double* Fw = t.F + iw * T_ * Ps.n;
            double* xpm = t.xp_mandel + iw * T_;
            tr_begin_part<D>(x + i * dd, g + i * dd, delta_tr[i], nullptr, nullptr, t, iw, Rw, C, mats, status, &B, x_unchanged, nlds);
            GABO_TICK(3);
            enum { PH_FD = 0, PH_PROP = 1, PH_REGRAD = 2 };
            int phase = PH_FD, it = 0, inner = 0;
            double* eg_it = egfd0;
            for (;;) {
                const double* ex = xfd;
                double* ev = t.val_fd + iw;
                double* eg = eg_it;
                bool do_eval = true;
                if (phase == PH_FD) {
                    ++inner;
                    if (!(it == 0 && x_unchanged && fd0_kept)) tcg_fd_point(w, iw, D, xfd, mats);
                    __syncthreads();
                    GABO_TICK(4);
                    eg_it = (it == 0) ? egfd0 : egfd;
                    eg = eg_it;
                    do_eval = !(it == 0 && x_unchanged);
                } else if (phase == PH_PROP) {
                    do_eval = !tr_build_proposal<D>(t, xp, iw, x_unchanged, step_cache, mats);
                    GABO_TICK(7);
                    ex = xpm;
                    ev = t.fx_prop + iw;
                    eg = lazy ? nullptr : t.eg_prop + iw * T_;
                } else {
                    ex = xpm;
                    ev = t.fx_prop + iw;
                    eg = t.eg_prop + iw * T_;
                }
                if (do_eval) acq_eval_any<D, METRIC>(ex, Ps, ev, eg, Fw, acq, dyn, status, iw + w.index_base);
                if (phase == PH_FD) {
                    __syncthreads();
                    GABO_TICK(5);
                    const bool running = tcg_step(w, iw, Rw, D, C, eg_it, 0, delta_cons, theta, kappa, mininner, it, mats);
                    __syncthreads();
                    GABO_TICK(6);
                    ++it;
                    if (!running || it >= maxinner) phase = PH_PROP;
                } else if (phase == PH_PROP) {
                    GABO_TICK(8);
                    __syncthreads();
                    if (rec != nullptr && rec_k < rec_cap) {          // (the iterate, its radius and the stop reason of the tCG run that made the proposal)
                        double* rr = rec + (rec_k * R + i) * (dd + 2);
                        for (int e = threadIdx.x; e < dd; e += 64) rr[e] = x[i * dd + e];
                        if (threadIdx.x == 0) {
                            rr[dd] = delta_tr[i];
                            rr[dd + 1] = (double)w.stop[iw];
                        }
                    }
                    ++rec_k;
                    inval = (B.strict && C > 0) ? builtin_infeasible<D>(xp, B, nlds) : false;
                    if (lazy && tr_would_accept(fx[i], t.fx_prop[iw], t.rhoden[iw], inval, rho_prime, rho_regularization)) phase = PH_REGRAD;
                    else break;
                } else {
                    __syncthreads();
                    break;
                }
            }
            last_inner = inner;
        }
#endif
        bool accepted = false;
        const bool still = tr_update_body(x + i * dd, fx + i, g + i * dd, ng + i, delta_tr + i, iters + i, inval, xp, t, iw, D, C, delta_bar,
                                          rho_prime, rho_regularization, mingradnorm, maxiter, mats, &accepted);
        GABO_TICK(9);
#ifndef GABO_TR_NO_FAST_FORWARD
        if (still && !accepted && last_inner == 1 && t.tcg.running[iw] == 0 && shortcuts != 0 && rec == nullptr && rho_prime < 0.25) {
            bool still_ff = true;
            tr_repeat_rejected(t.tcg, iw, Rw, D, C, mats + 4 * dd, delta_cons, delta_tr + i, iters + i, maxiter, &still_ff);
            if (!still_ff) break;
        }
#endif
        if (!still) break;
        cons_fresh = !accepted;
    }
    if (threadIdx.x == 0) active[i] = 0;
    if (own_finish) {
        __syncthreads();
        tr_finish_body<D>(S, x + i * dd, fx[i], iters[i]);
        if (S.status_host != nullptr && threadIdx.x == 0) {          // (mirror_status of spd_acq_kernel.hpp)
            const int e = __atomic_load_n(status, __ATOMIC_RELAXED);
            if (e != 0) {
                S.status_host[1] = __atomic_load_n(status + 1, __ATOMIC_RELAXED);
                S.status_host[0] = e;
            }
        }
    }
}

// The sweep's start of every restart as a launch of its own, in front of the solve (see the note in spd_tr_solve_kernel): one wave per restart,
// tr_start_body with the GP factors staged in LDS like the solve; the workspace slices it uses as scratch (xp_mandel, eg_prop, F) are the
// caller's global ones (the solve zeroes or ignores them).
template <int D, int METRIC>
__global__ __launch_bounds__(64) void spd_tr_start_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                          double* __restrict__ ng, double* __restrict__ delta_tr, uint8_t* __restrict__ active,
                                                          int64_t* __restrict__ iters, AcqParams P, void* wsbase, int64_t R, int C,
                                                          int* __restrict__ status, int stage_gp, TrStart S) {
    constexpr int dd = D * D;
    __shared__ AcqLds<D> acq;
    __shared__ __attribute__((aligned(16))) double mats[5 * dd + kJacobiScratch];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    if (S.skip_flag != nullptr && *S.skip_flag != 0) {          // (uniform over the launch)
        if (threadIdx.x == 0) {
            active[i] = 0;
            iters[i] = 0;
            fx[i] = __builtin_nan("");
        }
        return;
    }
    const AcqParams Ps = stage_gp_factors(P, dyn, stage_gp);
    TrWs t = tr_layout(wsbase, R, D, C, P.n);
    tr_start_body<D, METRIC>(S, x + i * dd, fx + i, g + i * dd, ng + i, Ps, t, i, acq, mats, dyn, status);
    if (threadIdx.x == 0) {
        delta_tr[i] = S.delta0;
        iters[i] = 0;
        active[i] = 1;
    }
}

// one translation unit per metric (spd_tr.hip, spd_tr_le.hip, spd_tr_frob.hip) so that the instantiations compile in parallel
template <int D, int METRIC>
static int launch_propose_one(const double* x, const double* g, const double* delta_tr, const uint8_t* active, const double* gc,
                              const double* fc, const AcqParams& P, void* ws, double* x_prop, int64_t r, int c, int neq,
                              double delta_cons, double theta, double kappa, int mininner, int maxinner, int* any_active, int* status,
                              hipStream_t st) {
    int stage_gp = 0;
    size_t lds = tr_dynamic_lds(P.n, r, &stage_gp, tr_factor_count(P));
    hipLaunchKernelGGL((spd_tr_propose_kernel<D, METRIC>), dim3((unsigned)r), dim3(64), lds, st, x, g, delta_tr, active, gc, fc, P, ws,
                       x_prop, r, c, neq, delta_cons, theta, kappa, mininner, maxinner, any_active, status, stage_gp);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

struct ProposeArgs {
    const double *x, *g, *delta_tr;
    const uint8_t* active;
    const double *gc, *fc;
    const AcqParams* P;
    void* ws;
    double* x_prop;
    int64_t r;
    int d, c, neq;
    double delta_cons, theta, kappa;
    int mininner, maxinner;
    int *any_active, *status;
    hipStream_t st;
};

template <int METRIC, int DMAX, int DMIN = 2>
static int dispatch_propose(const ProposeArgs& a) {
#define GABO_CASE(DD)                                                                                                                  \
    case DD:                                                                                                                           \
        if constexpr (DD <= DMAX && DD >= DMIN)                                                                                                      \
            return launch_propose_one<DD, METRIC>(a.x, a.g, a.delta_tr, a.active, a.gc, a.fc, *a.P, a.ws, a.x_prop, a.r, a.c, a.neq, \
                                                  a.delta_cons, a.theta, a.kappa, a.mininner, a.maxinner, a.any_active, a.status, a.st); \
        else                                                                                                                           \
            return GABO_ERR_DIM;
    switch (a.d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

struct SolveArgs {
    double *x, *fx, *g, *ng, *delta_tr;
    uint8_t* active;
    int64_t* iters;
    const AcqParams* P;
    BuiltinCons B;
    void* ws;
    int64_t r;
    int d;
    double delta_cons, theta, kappa;
    int mininner, maxinner;
    double delta_bar, rho_prime, rho_regularization, mingradnorm;
    int64_t maxiter;
    int* status;
    hipStream_t st;
    double* rec = nullptr;  // gabo_tr_solve_record: per-iteration record of this call, or null
    int64_t rec_cap = 0;
    int shortcuts = 1;      // 0: every iteration computes its proposal and the full evaluation (the environment variable GABO_TR_NO_SHORTCUTS: tests)
    TrStart start = {nullptr, 0, nullptr, 0.0, nullptr, nullptr, nullptr, nullptr};      // the sweep driver's start / end inside the launch (spd_sweep.hip)
};

// Round 5 history of two instantiations (tools/soak_tr.py found them; tools/repro_solve_fault.py walks the whole table): with the
// log-Euclidean evaluation's adjoint unrolled in every lane's registers (spd_acq_body.hpp, -DGABO_FROB_REGISTER_ADJOINT) the kernels of that
// surrogate at d = 7, 8 were 512-register functions with ~3700 vector and ~880 scalar registers spilled, and two of them were wrong AS COMPILED:
// the generic-workspace solve faulted on a null address at its first launch, the d = 8 propose kernel returned wrong proposals - while the
// LDS-resident solve of the same source was right.  The adjoint is now shared by the wave through LDS (bit-identical results, 233 registers
// at d = 8, no spills to speak of) and both instantiations are right again (the same table, the soak).  GABO_LE_MAX_GENERIC_DIM = 6 builds the
// library without them, as the first fix did; gabo_spd_tr_solve_supported / gabo_spd_tr_propose_supported answer for whatever was built.
#ifndef GABO_LE_MAX_GENERIC_DIM
#define GABO_LE_MAX_GENERIC_DIM 8
#endif
static constexpr bool solve_needs_lds_workspace(int metric, int d) { return metric == 1 && d > GABO_LE_MAX_GENERIC_DIM; }

// whether dispatch_solve would launch for this problem (the same sizing decisions)
static inline bool solve_supported(int metric, int64_t n, int64_t r, int d, int C, bool has_factors, size_t nested_bytes, int factors = 2) {
    int stage_gp = 0, ws_lds = 0;
    const size_t lds = tr_solve_dynamic_lds(n, r, d, C, &stage_gp, &ws_lds, nested_bytes, nullptr, factors);
    if (lds > 64 * 1024 || d < 2 || d > 8) return false;
    const bool lat = stage_gp && ws_lds && has_factors;
    return lat || !solve_needs_lds_workspace(metric, d);
}

template <int METRIC, int DMIN = 2, int DMAX = 8>
static int dispatch_solve(const SolveArgs& a) {
    int stage_gp = 0, ws_lds = 0, nested_off = 0;
    const size_t nested_bytes = builtin_has_kind(a.B, true) ? nested_extremes_lds_doubles(a.B.big_dim, a.d) * sizeof(double) : 0;
    size_t lds = tr_solve_dynamic_lds(a.P->n, a.r, a.d, a.B.n, &stage_gp, &ws_lds, nested_bytes, &nested_off, tr_factor_count(*a.P));
    if (lds > 64 * 1024) return GABO_ERR_ARG;
#ifdef GABO_TR_NO_LAT    /* A/B: the runtime-flag kernel everywhere */
    const bool lat = false;
#else
    const bool lat = stage_gp && ws_lds && a.P->linv && a.P->linv_t;
#endif
#define GABO_SOLVE_LAUNCH(DD, LAT_)                                                                                                \
    hipLaunchKernelGGL((spd_tr_solve_kernel<DD, METRIC, LAT_>), dim3((unsigned)a.r), dim3(64), lds, a.st, a.x, a.fx, a.g, a.ng, a.delta_tr, \
                       a.active, a.iters, *a.P, a.B, a.ws, a.r, a.delta_cons, a.theta, a.kappa, a.mininner, a.maxinner, a.delta_bar, \
                       a.rho_prime, a.rho_regularization, a.mingradnorm, a.maxiter, a.status, stage_gp, ws_lds, nested_off, a.shortcuts, a.rec, a.rec_cap, \
                       a.start)
#define GABO_CASE(DD)                                                                                                              \
    case DD:                                                                                                                       \
        if constexpr (DD >= DMIN && DD <= DMAX) {                                                                                  \
            if (a.start.raw_rows != nullptr) {                                                                                     \
                int sgp = 0;                                                                                                       \
                const size_t slds = tr_dynamic_lds(a.P->n, a.r, &sgp, tr_factor_count(*a.P));                                      \
                hipLaunchKernelGGL((spd_tr_start_kernel<DD, METRIC>), dim3((unsigned)a.r), dim3(64), slds, a.st, a.x, a.fx, a.g, a.ng, a.delta_tr, \
                                   a.active, a.iters, *a.P, a.ws, a.r, a.B.n, a.status, sgp, a.start);                             \
            }                                                                                                                      \
            if (lat) GABO_SOLVE_LAUNCH(DD, true);                                                                                  \
            else if constexpr (solve_needs_lds_workspace(METRIC, DD)) return GABO_ERR_DIM;                                         \
            else GABO_SOLVE_LAUNCH(DD, false);                                                                                     \
        } else {                                                                                                                   \
            return GABO_ERR_DIM;                                                                                                   \
        }                                                                                                                          \
        break;
    switch (a.d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8)
        default: return GABO_ERR_DIM;
    }
#undef GABO_CASE
#undef GABO_SOLVE_LAUNCH
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

// gabo_tr_solve_record (spd_tr.hip): hands the pending record buffer to the solve that is being launched and clears it
void tr_record_take(double** buffer, int64_t* capacity);

// defined in spd_tr.hip / spd_tr_le.hip / spd_tr_frob.hip / spd_tr_solve.hip / spd_tr_solve_le.hip
int solve_affine_invariant(const SolveArgs& a);
int solve_log_euclidean(const SolveArgs& a);
int propose_affine_invariant(const ProposeArgs& a);
int propose_affine_invariant_wide(const ProposeArgs& a);      // d = 9..12 (spd_tr_wide.hip)
int propose_log_euclidean(const ProposeArgs& a);
int propose_frobenius(const ProposeArgs& a);                  // spd_tr_frob.hip
int solve_frobenius(const SolveArgs& a);                      // spd_tr_solve_frob.hip

// what gabo_spd_tr_solve does behind its argument checks (spd_tr.hip); spd_sweep.hip calls it with `start` set.  *needs_zeroed_workspace (may be
// null): whether this problem runs with the caller's workspace (which must then be zeroed) instead of the block-private LDS copy.
int tr_solve_dispatch(const SolveArgs& a);
bool tr_solve_uses_global_workspace(const AcqParams& P, int64_t r, int d, int C, size_t nested_bytes);

}  // namespace gabo
