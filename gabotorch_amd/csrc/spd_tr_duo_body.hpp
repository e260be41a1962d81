// Single-launch trust-region solve with TWO waves per restart (round 6; affine-invariant surrogate, built-in eigenvalue bounds, latency regime).
//
// The one-wave kernel (spd_tr_solve_kernel, spd_tr_body.hpp) walks an accepted iteration as one dependent chain: tCG begin -> constraints at x -> FD point ->
// acquisition there -> tCG step -> proposal -> acquisition at the proposal -> update (138 k cycles at d = 5, n = 50; two evaluations of 40 k each).  In config 4
// 56 % of the iterations (gabo_spd_tr_two_waves_counters; per restart: tools/duo_times.py) tCG leaves in its FIRST step through the trust-region boundary or
// negative curvature, and that step does not depend on the Hessian-vector product at all: eta = tau delta_0 with tau from (Delta, <delta_0, delta_0>) and the
// linearised constraints (robust_trust_regions.py:500-512, constrained_trust_regions.py:583-640) - the product only DECIDES that the branch is taken.  So the
// block is two waves:
//   wave 0 (the tCG wave):       begin, FD point, acquisition at the FD point, tCG step(s), model decrease, update (and straight into the next begin)
//   wave 1 (the proposal wave):  the boundary step as tCG would take it, proposal x+ = L expm(eta~) L^T, acquisition at x+; then, while the other wave runs the
//                                update and the next begin, the eigen-pairs of the proposal it expects to be accepted (the next iterate's constraints)
// running side by side.  The tCG wave computes the speculated step too (scalars and one axpy) and compares its own eta~ with it element by element.  Equal: the
// proposal and its value are the ones the one-wave kernel computes (the same statements on the same operands: the same bits), and the iteration took
// max(...) instead of the sum (82 k cycles).  Not equal (an interior step, more tCG iterations): the tCG wave - which knows without waiting - finishes, then the
// proposal wave builds and evaluates the real proposal: the one-wave schedule.  A rejected proposal shrinks the radius and leaves x, g, delta_0 in place: the
// next boundary step is speculated the same way, and a run of rejected proposals whose first step does not change is applied as scalar updates
// (tr_repeat_rejected, spd_tr_body.hpp).
//
// Synchronisation: the device bodies this kernel shares with the one-wave kernels synchronise with __syncthreads(), which for a 64-thread block IS a wave-level
// fence (the compiler drops the s_barrier).  The translation unit of this kernel (spd_tr_solve_duo.hip) therefore defines __syncthreads() as that wave-level
// fence before including them, and the block-level barriers between the two waves are explicit duo_block_sync() calls, executed by both waves in the same
// sequence - per iteration B1 (begin done | eigen-pairs done), B2 (tCG done | speculated proposal evaluated), B4 after a miss (real proposal evaluated), B5 when a
// value-only evaluation is followed by its gradient; every condition around one is block-uniform (words written before the previous barrier).  Between two
// barriers a wave reads nothing the other writes: the tCG wave snapshots what the speculation needs (delta_0, the scalars) before B1, each wave has its own
// scratch (AcqLds, mats, 3 n doubles, the logm spill F), its own copy of the constraints' values and whitened gradients and of the speculated step.
// The block is dim3(64, 2): threadIdx.x stays the lane index the shared bodies use, threadIdx.y is the wave.
// -DGABO_DUO_COOP (off: measured 2-3 % slower, CHANGELOG round 6 item 11): the evaluations during which one wave would idle are shared by both.
#pragma once
#ifndef GABO_DUO_TU
#error "include from spd_tr_solve_duo.hip only (it redefines __syncthreads for the bodies it includes)"
#endif
#include "spd_tr_body.hpp"

namespace gabo {

static __device__ __forceinline__ void duo_block_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// lambda_max / lambda_min of x and their eigenvectors: the first half of builtin_constraints (kinds 0 / 1), same statements
template <int D>
__device__ __forceinline__ void duo_extremes(const double* __restrict__ x, double& lmax, double& lmin, double (&vmax)[D], double (&vmin)[D]) {
    double lam[D], v[D * D];
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

// ... and the second half: values and whitened gradients from the eigen-pairs and L = chol(x); fc_out: C doubles, gc_out: C x D^2
template <int D>
__device__ __forceinline__ void duo_cons_publish(const double* L, const BuiltinCons& B, double lmax, double lmin, const double (&vmax)[D],
                                                 const double (&vmin)[D], double* fc_out, double* gc_out) {
    constexpr int dd = D * D;
    for (int k = 0; k < B.n; ++k) {
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
            fc_out[k] = want_max ? B.bound[k] - best : best - B.bound[k];
            double* out = gc_out + (int64_t)k * dd;
            static_for<D>([&](auto rr) {
                static_for<D>([&](auto cc) { out[decltype(rr)::value * D + decltype(cc)::value] = sign * u[decltype(rr)::value] * u[decltype(cc)::value]; });
            });
        }
    }
}

// "The same step again" of tr_build_proposal: is sym(etaw) the step the proposal in the workspace was built from (x unchanged)?  Rewrites the cache.
template <int D>
__device__ __forceinline__ bool duo_same_step(const double* etaw, double* step_cache, bool x_unchanged) {
    constexpr int T = tri_size(D);
    constexpr int dd = D * D;
    if (step_cache == nullptr) return false;
    bool same = x_unchanged && step_cache[T] != 0.0;
    for (int e = threadIdx.x; e < dd; e += 64) {
        const int r = e / D, c = e - r * D;
        const double sym = 0.5 * (etaw[r * D + c] + etaw[c * D + r]);
        if (r >= c) {
            same = same && (sym == step_cache[tri(r, c)]);
        }
    }
    same = __builtin_amdgcn_ballot_w64(!same) == 0;
    __syncthreads();
    for (int e = threadIdx.x; e < dd; e += 64) {
        const int r = e / D, c = e - r * D;
        if (r >= c) step_cache[tri(r, c)] = 0.5 * (etaw[r * D + c] + etaw[c * D + r]);
    }
    if (threadIdx.x == 0) step_cache[T] = 1.0;
    __syncthreads();
    return same;
}

// x+ = L expm(eta~) L^T and its Mandel vector: the second half of tr_build_proposal (D <= 8), same statements
template <int D>
__device__ __forceinline__ void duo_proposal_from_eta(const double* chol, const double* etaw, double* __restrict__ x_prop, double* __restrict__ xpm,
                                                      double* mats) {
    constexpr int T = tri_size(D);
    constexpr int dd = D * D;
    double* M0 = mats;
    double* M1 = M0 + dd;
    double* M2 = M1 + dd;
    double* M3 = M2 + dd;
    lds_load(chol, M0, D);
    {
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
    }
    lds_congruence(M0, M3, M1, M2, D);
    lds_symmetrize(M1, M2, D);
    lds_store(M1, x_prop, D);
    for (int e = threadIdx.x; e < T; e += 64) {
        int k = 0;
        while (k + 1 < D && (k + 1) * D - (k + 1) * k / 2 <= e) ++k;
        int cc = e - (k * D - k * (k - 1) / 2);
        int r = cc + k;
        xpm[e] = (k == 0) ? M1[r * D + cc] : kSqrt2 * M1[r * D + cc];
    }
    __syncthreads();
}

// The step tCG takes when it leaves in its first iteration through the boundary branch of tcg_step_core (d_Hd <= 0 or the full step beyond the radius),
// from the state tcg_begin left: the same statements with the same operands (sc: the scalars as stored, zero: a stored 0.0 standing for eta~ and <grad c, eta>
// before the first step, gc / fc: this wave's copy of the constraints).  out: eta~ = 0 + tau delta~.
template <int D>
__device__ __forceinline__ void duo_boundary_step(const double* sc, const double* dl, const double* gc, const double* fc, int C, double delta_cons,
                                                  const double* zero, double* out) {
    constexpr int dd = D * D;
    const double Delta = sc[SC_DELTA], e_Pe = sc[SC_E_PE], e_Pd = sc[SC_E_PD], d_Pd = sc[SC_D_PD];
    const double dc2 = delta_cons * delta_cons;
    const double Delta2 = Delta * Delta;
    double fcl[kMaxCons], fpe[kMaxCons], fpd[kMaxCons];
    for (int k = 0; k < C; ++k) {
        fcl[k] = fc[k];
        fpe[k] = zero[0];
        fpd[k] = wave_dot(gc + (int64_t)k * dd, dl, dd);
    }
    double tau = (-e_Pd + __builtin_sqrt(e_Pd * e_Pd + d_Pd * (Delta2 - e_Pe))) / d_Pd;
    if (C > 0) {
        if (tau != tau) tau = 0.0;
        ConsStep cst = cons_step(tau, fcl, fpe, fpd, C, 0, dc2);
        if (cst.cin > dc2) tau = cst.tau;
    }
    const double step = tau;
    for (int e = threadIdx.x; e < dd; e += 64) out[e] = zero[0] + step * dl[e];
    __syncthreads();
}

// dynamic LDS of the two-wave kernel (bytes) for n training points, dimension d, C constraints; offsets in doubles
struct DuoLds {
    size_t scratch1, kinv, ws, f1, gc1, fc1, snap, coopw, bytes;
};
static __host__ __device__ inline DuoLds duo_lds_layout(int64_t n, int d, int C) {
    DuoLds l;
    const int dd = d * d;
    const size_t dv = (size_t)d * (d + 1) / 2;
    size_t off = 0;                                 // wave 0's 3 n doubles first
    off += 3 * n;
    l.scratch1 = off;  off += 3 * n;
    l.kinv = off;      off += (size_t)n * n;
    off = (off + 1) & ~(size_t)1;
    l.ws = off;        off += (tr_layout(nullptr, 1, d, C, n).bytes + 15) / 8;
    off = (off + 1) & ~(size_t)1;
    l.f1 = off;        off += dv * n;
    l.gc1 = off;       off += (size_t)(C > 0 ? C : 1) * dd;
    l.fc1 = off;       off += kMaxCons;
    l.snap = off;      off += dd + SC_COUNT + 2;      // delta_0~, the scalars, a stored zero
    l.coopw = off;     off += n;                      // the weights w_j of an evaluation shared by the two waves (acq_eval_values -> acq_eval_finish)
    l.bytes = off * sizeof(double);
    return l;
}

#ifdef GABO_DUO_TIMES
static __device__ long long g_duo_times[4 * 1024];
#endif

template <int D>
struct DuoStatic {
    static constexpr int dd = D * D;
    static constexpr int T = tri_size(D);
    AcqLds<D> acq[2];
    double mats[2][5 * dd + kJacobiScratch];
    double spec_eta[2][dd];       // the speculated step, one copy per wave
    double step_cache[T + 1];
    double cons_pub[2 + 2 * D];
    int flags[8];
};
enum { DF_STILL = 0, DF_ACCEPTED, DF_HIT, DF_INVAL, DF_COOP, DF_COOP2 };

template <int D>
__global__ __launch_bounds__(128) void spd_tr_solve_duo_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                               double* __restrict__ ng, double* __restrict__ delta_tr,
                                                               uint8_t* __restrict__ active, int64_t* __restrict__ iters, AcqParams P,
                                                               BuiltinCons B, int64_t R, double delta_cons, double theta, double kappa,
                                                               int mininner, int maxinner, double delta_bar, double rho_prime,
                                                               double rho_regularization, double mingradnorm, int64_t maxiter,
                                                               int* __restrict__ status, int shortcuts, double* __restrict__ rec, int64_t rec_cap,
                                                               TrStart S, int* __restrict__ counters) {      // counters: {hits, misses} or null
    static_assert(D <= 8, "register eigen-solvers");
    constexpr int dd = D * D;
    constexpr int T = tri_size(D);
    __shared__ DuoStatic<D> sh;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const int lane = threadIdx.x;
#ifdef GABO_DUO_TIMES       /* development (tools/duo_times.py): cycles, iterations, hits and misses of every restart */
    const long long t_start = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int64_t i = blockIdx.x;
    const bool own_finish = S.res_rows != nullptr;
    if (active[i] == 0) {          // (block-uniform)
        if (own_finish && wv == 0) tr_finish_body<D>(S, x + i * dd, fx[i], iters[i]);
        return;
    }
    const int C = B.n;
    const DuoLds lay = duo_lds_layout(P.n, D, C);
    double* dynw = wv == 0 ? dyn : dyn + lay.scratch1;        // this wave's 3 n doubles
    double* gl = dyn + lay.kinv;
    {   // the symmetric inverse of the Gram matrix, staged once by both waves
        const int64_t nn = P.n * P.n;
        for (int64_t e = lane + 64 * wv; e < nn; e += 128) gl[e] = P.linv[e];
    }
    AcqParams Ps = P;
    Ps.linv = gl;
    Ps.linv_t = gl;
    // the per-restart workspace of the one-wave kernel, in LDS
    char* wsbase = reinterpret_cast<char*>(dyn + lay.ws);
    TrWs t = tr_layout(wsbase, 1, D, C, P.n);
    for (size_t e = lane + 64 * wv; e < t.bytes / sizeof(double); e += 128) reinterpret_cast<double*>(wsbase)[e] = 0.0;
    t.tcg.index_base = i;
    const TcgWs& w = t.tcg;
    double* F1 = dyn + lay.f1;
    double* gc1 = dyn + lay.gc1;
    double* fc1 = dyn + lay.fc1;
    double* snap_dl = dyn + lay.snap;
    double* snap_sc = snap_dl + dd;
    double* zero = snap_sc + SC_COUNT;
    double* coopw = dyn + lay.coopw;
    double* mats = sh.mats[wv];
    AcqLds<D>& acq = sh.acq[wv];
    if (wv == 0 && lane == 0) {
        sh.step_cache[T] = 0.0;
        zero[0] = 0.0;
    }
    double* const step_cache = shortcuts != 0 ? sh.step_cache : nullptr;
    double* xp = t.xp_mat;
    double* xpm = t.xp_mandel;
    const double* xi = x + i * dd;
    duo_block_sync();
    int n_hit = 0, n_miss = 0, n_skip = 0;
    // Barrier sequence of BOTH waves, per iteration: B1, B2, [B4 after a miss], [B5 when a value-only evaluation is followed by the gradient]; every
    // bracket is decided from words the previous barrier ordered.  The iteration of the tCG wave ends with the update and runs straight into the next
    // begin; the proposal wave spends that time on the eigen-pairs of the proposal it expects to be accepted (the next iterate's constraints).
    if (wv == 0) {
        // ================= the tCG wave
        bool x_unchanged = false;          // the previous proposal was rejected: x, g, the constraints stand
        int last_inner = 0;
        int64_t rec_k = rec != nullptr ? iters[i] : 0;
        for (;;) {
            const bool lazy = x_unchanged && shortcuts != 0;
            const bool fd0_kept = last_inner == 1;
            const double fx_now = fx[i];
            GABO_TICK(20);
            tr_begin_part<D>(xi, g + i * dd, delta_tr[i], nullptr, nullptr, t, 0, 1, C, mats, status, nullptr, x_unchanged, nullptr);
            for (int e = lane; e < dd; e += 64) snap_dl[e] = w.delta_w[e];
            if (lane < SC_COUNT) snap_sc[lane] = w.scal[lane];
            GABO_TICK(21);
            duo_block_sync();                                           // B1
            GABO_TICK(22);
            if (C > 0 && !x_unchanged) {          // the constraints' values and whitened gradients from the other wave's eigen-pairs: this wave's copy
                double vmax[D], vmin[D];
                const double lmax = sh.cons_pub[0], lmin = sh.cons_pub[1];
                static_for<D>([&](auto rr) {
                    vmax[decltype(rr)::value] = sh.cons_pub[2 + decltype(rr)::value];
                    vmin[decltype(rr)::value] = sh.cons_pub[2 + D + decltype(rr)::value];
                });
                duo_cons_publish<D>(w.chol, B, lmax, lmin, vmax, vmin, w.fc, w.gc_w);
                __syncthreads();
            }
            // the step the other wave is speculating on, here for the comparison alone (so that a miss is known without waiting for that wave)
            duo_boundary_step<D>(snap_sc, snap_dl, w.gc_w, w.fc, C, delta_cons, zero, sh.spec_eta[0]);
            GABO_TICK(23);
            int inner = 0;
            bool hit = false;
            for (int it = 0; it < maxinner; ++it) {          // the loop of tr_propose_body
                ++inner;
                if (!(it == 0 && x_unchanged && fd0_kept)) tcg_fd_point(w, 0, D, t.x_fd, mats);
                __syncthreads();
                double* eg_it = (it == 0) ? t.eg_fd0 : t.eg_fd;
#ifdef GABO_DUO_COOP
                if (it == 0) {
                    if (!x_unchanged) acq_eval<D>(t.x_fd, Ps, t.val_fd, eg_it, t.F, acq, dynw, status, i);
                } else {
                    // the speculation has missed and the other wave has nothing to do: the evaluation is shared with it (acq_eval_values there, the
                    // eigenvectors and the gradient here)
                    if (lane == 0) sh.flags[DF_COOP] = 1;
                    duo_block_sync();                                   // Ba: the FD point is published, the helper starts
                    acq_eval_vectors<D>(t.x_fd, Ps, t.F, acq, status, i);
                    duo_block_sync();                                   // Bc: the weights are there
                    acq_eval_finish<D>(Ps, eg_it, t.F, coopw, acq);
                }
#else
                if (!(it == 0 && x_unchanged)) acq_eval<D>(t.x_fd, Ps, t.val_fd, eg_it, t.F, acq, dynw, status, i);
#endif
                __syncthreads();
                const bool running = tcg_step(w, 0, 1, D, C, eg_it, 0, delta_cons, theta, kappa, mininner, it, mats);
                __syncthreads();
                if (it == 0) {          // did tCG stop here, with the speculated step?
                    bool eq = true;
                    for (int e = lane; e < dd; e += 64) eq = eq && (w.eta_w[e] == sh.spec_eta[0][e]);
                    hit = (!running || maxinner <= 1) && (__builtin_amdgcn_ballot_w64(!eq) == 0);
                    if (hit) break;
                }
                if (!running) break;
            }
            n_hit += hit ? 1 : 0;
            n_miss += hit ? 0 : 1;
            // the model decrease -<g, eta> - 1/2 <eta, H eta>
            const double ge = wave_dot(w.g_w, w.eta_w, dd);
            const double ehe = wave_dot(w.eta_w, w.heta_w, dd);
            if (lane == 0) {
                t.rhoden[0] = -ge - 0.5 * ehe;
                sh.flags[DF_HIT] = hit ? 1 : 0;
                sh.flags[DF_COOP] = 0;          // (no further evaluation to share: the barrier below ends the helper's loop)
            }
            if (rec != nullptr && rec_k < rec_cap) {
                double* rr = rec + (rec_k * R + i) * (dd + 2);
                for (int e = lane; e < dd; e += 64) rr[e] = xi[e];
                if (lane == 0) {
                    rr[dd] = delta_tr[i];
                    rr[dd + 1] = (double)w.stop[0];
                }
            }
            ++rec_k;
            GABO_TICK(24);
            duo_block_sync();                                           // B2
            GABO_TICK(25);
#ifdef GABO_DUO_COOP
            // after a miss the other wave builds and evaluates the real proposal, and after a value-only evaluation that is going to be accepted it evaluates
            // the gradient: this wave takes the value half of those evaluations (one site for both: trip 0 / trip 1)
            bool inval = false, accept_pred = false;
            for (int trip = 0; trip < 2; ++trip) {
                bool help;
                if (trip == 0) {
                    if (hit) {
                        help = false;
                    } else {
                        duo_block_sync();                               // Bp: the proposal is built
                        help = sh.flags[DF_COOP2] != 0;
                    }
                } else {
                    inval = sh.flags[DF_INVAL] != 0;
                    accept_pred = tr_would_accept(fx_now, t.fx_prop[0], t.rhoden[0], inval, rho_prime, rho_regularization);
                    help = lazy && accept_pred;
                    if (!help) break;
                }
                if (help) {
                    acq_eval_values<D>(xpm, Ps, t.fx_prop, coopw, acq, dynw, status, i);
                    duo_block_sync();                                   // Bc
                }
                if (trip == 0) {
                    if (!hit) duo_block_sync();                         // B4
                } else {
                    duo_block_sync();                                   // B5
                }
            }
            GABO_TICK(26);
#else
            if (!hit) duo_block_sync();                                 // B4
            GABO_TICK(26);
            const bool inval = sh.flags[DF_INVAL] != 0;
            // (the verdict as the other wave computes it: it decides there whether the gradient is evaluated after a value-only evaluation and whether the
            // proposal's eigen-pairs are prepared for the next iteration)
            const bool accept_pred = tr_would_accept(fx_now, t.fx_prop[0], t.rhoden[0], inval, rho_prime, rho_regularization);
            if (lazy && accept_pred) duo_block_sync();                  // B5
#endif
            bool accepted = false;
            bool still = tr_update_body(x + i * dd, fx + i, g + i * dd, ng + i, delta_tr + i, iters + i, inval, xp, t, 0, D, C, delta_bar,
                                        rho_prime, rho_regularization, mingradnorm, maxiter, mats, &accepted);
#ifndef GABO_TR_NO_FAST_FORWARD
            // (a run of rejected proposals with the same first tCG step: applied as scalar updates, see tr_repeat_rejected)
            if (still && !accepted && inner == 1 && w.running[0] == 0 && shortcuts != 0 && rec == nullptr && rho_prime < 0.25)
                n_skip += tr_repeat_rejected(w, 0, 1, D, C, mats + 4 * dd, delta_cons, delta_tr + i, iters + i, maxiter, &still);
#endif
            if (lane == 0) {
                sh.flags[DF_STILL] = still ? 1 : 0;
                sh.flags[DF_ACCEPTED] = accepted ? 1 : 0;
                // tr_would_accept and tr_update_body are the same statements; should they ever disagree the other wave has prepared the wrong constraints:
                // reported as a failed launch rather than computed with
                if (accepted != accept_pred && status != nullptr && atomicCAS(status, 0, GABO_ERR_LAUNCH) == 0) status[1] = (int)i;
            }
            __syncthreads();
            GABO_TICK(27);
            if (!still) {
                duo_block_sync();                                       // the B1 the other wave is waiting at
                break;
            }
            x_unchanged = !accepted;
            last_inner = inner;
        }
    } else {
        // ================= the proposal wave
        bool first = true;
        bool accept_pred = false;          // the verdict of the update of the previous iteration, as this wave computed it (tr_would_accept)
        for (;;) {
            // eigen-pairs of the next iterate for its constraints: the start, or the proposal that is being accepted (the update copies it into x)
            GABO_TICK(30);
            if (C > 0 && (first || accept_pred)) {
                double lmax, lmin, vmax[D], vmin[D];
                duo_extremes<D>(first ? xi : xp, lmax, lmin, vmax, vmin);
                if (lane == 0) {
                    sh.cons_pub[0] = lmax;
                    sh.cons_pub[1] = lmin;
                    static_for<D>([&](auto rr) {
                        sh.cons_pub[2 + decltype(rr)::value] = vmax[decltype(rr)::value];
                        sh.cons_pub[2 + D + decltype(rr)::value] = vmin[decltype(rr)::value];
                    });
                }
                __syncthreads();
            }
            GABO_TICK(31);
            duo_block_sync();                                           // B1
            GABO_TICK(32);
            if (!first && sh.flags[DF_STILL] == 0) break;
            const bool x_unchanged = !first && sh.flags[DF_ACCEPTED] == 0;
            const bool lazy = x_unchanged && shortcuts != 0;
            const double fx_now = fx[i];
            first = false;
            if (C > 0 && !x_unchanged) {
                double vmax[D], vmin[D];
                const double lmax = sh.cons_pub[0], lmin = sh.cons_pub[1];
                static_for<D>([&](auto rr) {
                    vmax[decltype(rr)::value] = sh.cons_pub[2 + decltype(rr)::value];
                    vmin[decltype(rr)::value] = sh.cons_pub[2 + D + decltype(rr)::value];
                });
                duo_cons_publish<D>(w.chol, B, lmax, lmin, vmax, vmin, fc1, gc1);
                __syncthreads();
            }
            // one evaluation site, three uses: the speculated proposal, the real one after a miss, the gradient after a value-only evaluation whose
            // proposal is going to be accepted
            duo_boundary_step<D>(snap_sc, snap_dl, gc1, fc1, C, delta_cons, zero, sh.spec_eta[1]);
            GABO_TICK(33);
            enum { PH_SPEC = 0, PH_REAL = 1, PH_REGRAD = 2 };
            int phase = PH_SPEC;
            bool inval = false;
            for (;;) {
                bool do_eval = true;
                double* gout = t.eg_prop;
                if (phase != PH_REGRAD) {
                    const double* src = phase == PH_SPEC ? sh.spec_eta[1] : w.eta_w;
                    do_eval = !duo_same_step<D>(src, step_cache, x_unchanged);
                    if (do_eval) duo_proposal_from_eta<D>(w.chol, src, xp, xpm, mats);
                    gout = lazy ? nullptr : t.eg_prop;
                    GABO_TICK(34);
                }
#ifdef GABO_DUO_COOP
                // the real proposal after a miss and the gradient after a value-only evaluation are evaluated WITH the other wave (it has nothing else to
                // do then): the eigenvectors and the gradient here, the value half there.  The speculated proposal is this wave's alone.
                bool shared = phase != PH_SPEC && do_eval && gout != nullptr;
                if (phase == PH_REAL) {
                    if (lane == 0) sh.flags[DF_COOP2] = shared ? 1 : 0;
                    duo_block_sync();                                   // Bp
                }
                if (shared) {
                    acq_eval_vectors<D>(xpm, Ps, F1, acq, status, i);
                    duo_block_sync();                                   // Bc
                    acq_eval_finish<D>(Ps, gout, F1, coopw, acq);
                    __syncthreads();
                } else if (do_eval) {
                    acq_eval<D>(xpm, Ps, t.fx_prop, gout, F1, acq, dynw, status, i);
                    __syncthreads();
                }
#else
                if (do_eval) {
                    acq_eval<D>(xpm, Ps, t.fx_prop, gout, F1, acq, dynw, status, i);
                    __syncthreads();
                }
#endif
                if (phase == PH_REGRAD) {
                    duo_block_sync();                                   // B5
                    break;
                }
                // feasibility of the proposal (strict variant)
                inval = (B.strict && C > 0) ? builtin_infeasible<D>(xp, B, nullptr) : false;
                if (lane == 0) sh.flags[DF_INVAL] = inval ? 1 : 0;
                GABO_TICK(35);
#ifdef GABO_DUO_COOP
                if (phase == PH_SPEC) {
                    // B2, or before it the barriers of the evaluations the tCG wave shares with this one while it runs further tCG iterations
                    for (;;) {
                        duo_block_sync();                               // Ba / B2
                        if (sh.flags[DF_COOP] == 0) break;
                        acq_eval_values<D>(t.x_fd, Ps, t.val_fd, coopw, acq, dynw, status, i);
                        duo_block_sync();                               // Bc
                    }
                } else {
                    duo_block_sync();                                   // B4
                }
#else
                duo_block_sync();                                       // B2 (speculated proposal) / B4 (real one)
#endif
                GABO_TICK(36);
                if (phase == PH_SPEC && sh.flags[DF_HIT] == 0) {
                    phase = PH_REAL;
                    continue;
                }
                accept_pred = tr_would_accept(fx_now, t.fx_prop[0], t.rhoden[0], inval, rho_prime, rho_regularization);
                if (lazy && accept_pred) {
                    phase = PH_REGRAD;
                    continue;
                }
                break;
            }
        }
    }
    if (wv != 0) return;
#ifdef GABO_DUO_TIMES
    if (lane == 0 && i < 1024) {
        g_duo_times[4 * i] = (long long)__builtin_amdgcn_s_memtime() - t_start;
        g_duo_times[4 * i + 1] = iters[i];
        g_duo_times[4 * i + 2] = n_hit;
        g_duo_times[4 * i + 3] = n_miss + 1000 * n_skip;
    }
#endif
    if (lane == 0) {
        active[i] = 0;
        if (counters != nullptr) {
            atomicAdd(counters, n_hit);
            atomicAdd(counters + 1, n_miss);
        }
    }
    if (own_finish) {
        __syncthreads();
        tr_finish_body<D>(S, x + i * dd, fx[i], iters[i]);
        if (S.status_host != nullptr && lane == 0) {
            const int e = __atomic_load_n(status, __ATOMIC_RELAXED);
            if (e != 0) {
                S.status_host[1] = __atomic_load_n(status + 1, __ATOMIC_RELAXED);
                S.status_host[0] = e;
            }
        }
    }
}

}  // namespace gabo
