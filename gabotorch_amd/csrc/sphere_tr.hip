// Acquisition maximisation on the sphere S^(dim-1), R restarts, one wave per restart - the sphere twin of spd_acq.hip / spd_tr.hip:
//   gabo_sphere_acq_eval    acquisition value and Euclidean gradient at R points in one launch: kernel strip
//                           k_j = f(acos(clamp <x, X_j>)) (kernels_sphere.py:71-94,118-134, sphere_utils_torch.py:12-55), exact-GP
//                           posterior + EI / posterior mean ([3P], as gp_acquisition.hip), gradient sum_j w_j f'(c_j) X_j
//   gabo_sphere_tr_propose  truncated CG with the finite-difference Hessian (robust_trust_regions.py:417-570,
//                           constrained_trust_regions.py:530-732, approximate_hessian.py:11-62) on pymanopt's Sphere geometry
//                           ([3P], SURVEY App. B: inner = dot, proj(x, u) = u - <x, u> x, retr = normalise(x + u),
//                           transp(x1, x2, u) = proj(x2, u)), then the proposal and the acquisition at the proposal
//   gabo_sphere_tr_update   rho test and state update (robust_trust_regions.py:236-330)
//   gabo_sphere_tr_solve    the whole solve in one launch when there are no constraints
// The scalar logic of the tCG iteration is tcg_step_core of spd_tcg_body.hpp (shared with the SPD kernels).
#include "spd_tcg_body.hpp"

namespace gabo {

using SphAcq = gabo_sphere_acq_params;

struct SphWs {
    double *g, *eta, *heta, *r, *delta, *x_fd, *eg_fd, *hd0, *x_prop, *eg_prop, *gc, *scal, *fc, *fcg_pe, *val_fd, *fx_prop, *rhoden;
    int *stop, *running;
    size_t bytes;
};

static __host__ __device__ inline SphWs sph_layout(void* base, int64_t R, int dim, int C) {
    SphWs w;
    double* p = (double*)base;
    const int64_t m = R * dim;
    w.g = p;        p += m;
    w.eta = p;      p += m;
    w.heta = p;     p += m;
    w.r = p;        p += m;
    w.delta = p;    p += m;
    w.x_fd = p;     p += m;
    w.eg_fd = p;    p += m;
    w.hd0 = p;      p += m;       // Hessian-vector product of the FIRST tCG direction at the current x (kept while x does not move)
    w.x_prop = p;   p += m;
    w.eg_prop = p;  p += m;
    w.gc = p;       p += (int64_t)C * m;
    w.scal = p;     p += R * SC_COUNT;
    w.fc = p;       p += R * (C > 0 ? C : 1);
    w.fcg_pe = p;   p += R * (C > 0 ? C : 1);
    w.val_fd = p;   p += R;
    w.fx_prop = p;  p += R;
    w.rhoden = p;   p += R;
    int* q = (int*)p;
    w.stop = q;     q += R;
    w.running = q;  q += R;
    w.bytes = (size_t)((char*)q - (char*)base);
    return w;
}

// sum_{j < n} col[j * n] * x[j]: one output of a matrix-vector product with an n x n matrix per lane (strided_dot of spd_acq_body.hpp).  The GP factors
// L^-1 and L^-T are stored dense with exact zeros outside their triangle, so the sum runs over ALL j: a uniform trip count, eight loads issued before the
// FMAs that use them, four partial sums.  (Rounds 2-6a walked the triangle with a lane-dependent bound, one dependent load per term from L2: the two - in
// the Hessian-vector evaluation four - such loops were most of an evaluation.)
static __device__ __forceinline__ double sph_col_dot(const double* __restrict__ col, const double* __restrict__ x, int n, int stride = 0) {
    if (stride == 0) stride = n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        double av[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = col[(j + u) * stride];
            xv[u] = x[j + u];
        }
        s0 = __builtin_fma(av[0], xv[0], s0); s1 = __builtin_fma(av[1], xv[1], s1);
        s2 = __builtin_fma(av[2], xv[2], s2); s3 = __builtin_fma(av[3], xv[3], s3);
        s0 = __builtin_fma(av[4], xv[4], s0); s1 = __builtin_fma(av[5], xv[5], s1);
        s2 = __builtin_fma(av[6], xv[6], s2); s3 = __builtin_fma(av[7], xv[7], s3);
    }
    for (; j < n; ++j) s0 = __builtin_fma(col[j * stride], x[j], s0);
    return (s0 + s1) + (s2 + s3);
}

// value and (grad != nullptr) Euclidean gradient of out_sign * acquisition at x (dim doubles, global or LDS).  dyn: 3 n doubles of LDS.
// P.linv == P.linv_t: the caller handed over the symmetric A = (outputscale K + noise I)^-1 (gabo_gp_factor's kinv) for both: one matrix-vector
// product per evaluation instead of two (as the SPD evaluations, spd_acq_body.hpp).
static __device__ __forceinline__ void sph_acq_eval(const double* __restrict__ x, const SphAcq& P, double* __restrict__ value_out,
                                    double* __restrict__ grad_out, double* dyn) {
    const int64_t n = P.n;
    const int dim = P.dim;
    double* ks = dyn;
    double* kd = ks + n;      // f'(c_j), then the weights w_j
    double* vv = kd + n;
    const int lane = threadIdx.x;
    const int mode = P.flags & GABO_OUT_MASK;
    const bool sym = P.linv != nullptr && P.linv == P.linv_t;
    const double lo = -1.0 + 1e-15, hi = 1.0 - 1e-15;       // sphere_utils_torch.py:53
    for (int64_t j = lane; j < n; j += 64) {
        double ip = 0.0;
        for (int k = 0; k < dim; ++k) ip = __builtin_fma(x[k], P.train_t[(int64_t)k * n + j], ip);
        const bool inside = ip >= lo && ip <= hi;
        const double c = ip < lo ? lo : (ip > hi ? hi : ip);
        const double th = acos(c);
        const double t1 = -1.0 / __builtin_sqrt((1.0 - c) * (1.0 + c));       // theta'(c)
        double kj, dk;
        if (mode == GABO_OUT_GAUSSIAN) {
            kj = exp(-((th * th) * P.beta));
            dk = kj * (-2.0 * P.beta * th * t1);
        } else {
            kj = exp(-(th * P.beta));
            dk = kj * (-P.beta * t1);
        }
        ks[j] = P.outputscale * kj;
        kd[j] = inside ? dk : 0.0;                                              // clamp passes no gradient outside
    }
    __syncthreads();
    double part = 0.0;
    for (int64_t j = lane; j < n; j += 64) part = __builtin_fma(ks[j], P.alpha[j], part);
    const double mean = P.mean + wave_sum(part);
    const double sgn = P.maximize ? 1.0 : -1.0;
    double g_mean, g_var = 0.0;
    if (P.kind == GABO_ACQ_POSTERIOR_MEAN) {
        if (lane == 0) *value_out = P.out_sign * sgn * mean;
        g_mean = sgn;
    } else {
        part = 0.0;
        for (int64_t r = lane; r < n; r += 64) {
            const double a = sph_col_dot((sym ? P.linv : P.linv_t) + r, ks, (int)n);      // (A ks)_r, or (L^-1 ks)_r = sum_j L^-T[j][r] ks[j]
            vv[r] = a;
            part = sym ? __builtin_fma(ks[r], a, part) : __builtin_fma(a, a, part);
        }
        const double var = P.outputscale * P.kxx - wave_sum(part);
        const bool clamped = !(var > 1e-9);
        const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
        const double u = sgn * (mean - P.best_f) / sigma;
        const double pdf = exp(-0.5 * u * u) * 0.3989422804014327;
        const double cdf = 0.5 * (1.0 + erf(u * 0.7071067811865476));
        if (lane == 0) *value_out = P.out_sign * sigma * (pdf + u * cdf);
        g_mean = sgn * cdf;
        g_var = clamped ? 0.0 : 0.5 * pdf / sigma;
    }
    if (grad_out == nullptr) return;
    __syncthreads();
    for (int64_t j = lane; j < n; j += 64) {
        double ws = 0.0;
        if (P.kind != GABO_ACQ_POSTERIOR_MEAN) ws = sym ? vv[j] : sph_col_dot(P.linv + j, vv, (int)n);      // (L^-T v)_j = sum_r L^-1[r][j] v[r]
        const double gk = P.out_sign * P.outputscale * (g_mean * P.alpha[j] - 2.0 * g_var * ws);
        kd[j] = gk * kd[j];                                                    // w_j = d/dc_j
    }
    __syncthreads();
    for (int k = lane; k < dim; k += 64) grad_out[k] = sph_col_dot(P.train + k, kd, (int)n, dim);      // sum_j w_j X_j[k]
}

// Exact Euclidean Hessian-vector product of out_sign * acquisition at x along u (what double backward through the sphere kernel,
// the GP posterior and EI gives the reference: pymanopt_addons/tools/autodiff/_pytorch.py:103-116), together with the gradient:
//   c_j = <x, X_j>, s_j = <u, X_j>;  ks = os f(c), d = os f'(c) s;  q = K^-1 ks, qd = K^-1 d;  mu' = alpha.d, var' = -2 q.d
//   G_j = A_mu alpha_j - 2 A_v q_j,  G'_j = (A_mumu mu' + A_muv var') alpha_j - 2 (A_muv mu' + A_vv var') q_j - 2 A_v qd_j
//   egrad = sum_j G_j os f'(c_j) X_j,   ehess u = sum_j [G'_j os f'(c_j) + G_j os f''(c_j) s_j] X_j
// dyn: 7 n doubles of LDS.
static __device__ __forceinline__ void sph_acq_hess(const double* __restrict__ x, const double* __restrict__ u, const SphAcq& P,
                                    double* __restrict__ egrad_out, double* __restrict__ ehess_out, double* dyn) {
    const int64_t n = P.n;
    const int dim = P.dim;
    double* ks = dyn;          // os f(c)
    double* f1 = ks + n;       // os f'(c)  (0 outside the clamp)
    double* f2 = f1 + n;       // os f''(c) (0 outside the clamp)
    double* sj = f2 + n;       // <u, X_j>
    double* vv = sj + n;       // scratch: L^-1 (.)
    double* q = vv + n;        // K^-1 ks
    double* qd = q + n;        // K^-1 d
    const int lane = threadIdx.x;
    const int mode = P.flags & GABO_OUT_MASK;
    const double lo = -1.0 + 1e-15, hi = 1.0 - 1e-15;
    for (int64_t j = lane; j < n; j += 64) {
        double ip = 0.0, su = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double xt = P.train_t[(int64_t)k * n + j];
            ip = __builtin_fma(x[k], xt, ip);
            su = __builtin_fma(u[k], xt, su);
        }
        const bool inside = ip >= lo && ip <= hi;
        const double c = ip < lo ? lo : (ip > hi ? hi : ip);
        const double th = acos(c);
        const double om = (1.0 - c) * (1.0 + c);
        const double t1 = -1.0 / __builtin_sqrt(om);
        const double t2 = c * t1 / om;
        double kj, d1, d2;
        if (mode == GABO_OUT_GAUSSIAN) {
            kj = exp(-((th * th) * P.beta));
            const double a = -2.0 * P.beta * th * t1;
            d1 = kj * a;
            d2 = kj * (a * a - 2.0 * P.beta * (t1 * t1 + th * t2));
        } else {
            kj = exp(-(th * P.beta));
            const double a = -P.beta * t1;
            d1 = kj * a;
            d2 = kj * (a * a - P.beta * t2);
        }
        ks[j] = P.outputscale * kj;
        f1[j] = inside ? P.outputscale * d1 : 0.0;
        f2[j] = inside ? P.outputscale * d2 : 0.0;
        sj[j] = su;
    }
    __syncthreads();
    const double sgn = P.maximize ? 1.0 : -1.0;
    double A_mu, A_v = 0.0, A_mumu = 0.0, A_muv = 0.0, A_vv = 0.0, mud = 0.0, vard = 0.0;
    double part = 0.0, pd = 0.0;
    for (int64_t j = lane; j < n; j += 64) {
        part = __builtin_fma(ks[j], P.alpha[j], part);
        pd = __builtin_fma(f1[j] * sj[j], P.alpha[j], pd);
    }
    const double mean = P.mean + wave_sum(part);
    mud = wave_sum(pd);
    if (P.kind == GABO_ACQ_POSTERIOR_MEAN) {
        A_mu = sgn;
        for (int64_t j = lane; j < n; j += 64) { q[j] = 0.0; qd[j] = 0.0; }
    } else {
        // q = K^-1 ks, qd = K^-1 d with d_j = os f'(c_j) s_j
        double var;
        if (P.linv == P.linv_t) {
            // (the symmetric inverse A for both: two matrix-vector products instead of four)
            double pv = 0.0;
            for (int64_t j = lane; j < n; j += 64) {
                const double a = sph_col_dot(P.linv + j, ks, (int)n);
                q[j] = a;
                pv = __builtin_fma(ks[j], a, pv);
                vv[j] = f1[j] * sj[j];
            }
            var = P.outputscale * P.kxx - wave_sum(pv);
            __syncthreads();
            double pq = 0.0;
            for (int64_t j = lane; j < n; j += 64) {
                qd[j] = sph_col_dot(P.linv + j, vv, (int)n);
                pq = __builtin_fma(q[j], vv[j], pq);
            }
            vard = -2.0 * wave_sum(pq);
        } else {
            // q = L^-T (L^-1 ks), qd = L^-T (L^-1 d)
            double pv = 0.0;
            for (int64_t r = lane; r < n; r += 64) {
                const double a = sph_col_dot(P.linv_t + r, ks, (int)n);
                vv[r] = a;
                pv = __builtin_fma(a, a, pv);
            }
            var = P.outputscale * P.kxx - wave_sum(pv);
            __syncthreads();
            for (int64_t j = lane; j < n; j += 64) q[j] = sph_col_dot(P.linv + j, vv, (int)n);
            __syncthreads();
            for (int64_t j = lane; j < n; j += 64) qd[j] = f1[j] * sj[j];          // (d, staged where qd will be)
            __syncthreads();
            for (int64_t r = lane; r < n; r += 64) vv[r] = sph_col_dot(P.linv_t + r, qd, (int)n);
            __syncthreads();
            double pq = 0.0;
            for (int64_t j = lane; j < n; j += 64) {
                const double dj = f1[j] * sj[j];
                qd[j] = sph_col_dot(P.linv + j, vv, (int)n);
                pq = __builtin_fma(q[j], dj, pq);
            }
            vard = -2.0 * wave_sum(pq);
        }
        const bool clamped = !(var > 1e-9);
        const double sigma = __builtin_sqrt(clamped ? 1e-9 : var);
        const double uu = sgn * (mean - P.best_f) / sigma;
        const double pdf = exp(-0.5 * uu * uu) * 0.3989422804014327;
        const double cdf = 0.5 * (1.0 + erf(uu * 0.7071067811865476));
        A_mu = sgn * cdf;
        A_mumu = pdf / sigma;                                  // d/dmu (s Phi(u)) = phi / sigma
        if (!clamped) {
            A_v = 0.5 * pdf / sigma;
            A_muv = -sgn * pdf * uu / (2.0 * var);
            A_vv = pdf * (uu * uu - 1.0) / (4.0 * sigma * sigma * sigma);
        }
    }
    __syncthreads();
    const double cm = A_mumu * mud + A_muv * vard;
    const double cv = A_muv * mud + A_vv * vard;
    // weights on X_j: wg_j (gradient) in ks, wh_j (Hessian-vector) in vv
    for (int64_t j = lane; j < n; j += 64) {
        const double G = A_mu * P.alpha[j] - 2.0 * A_v * q[j];
        const double Gd = cm * P.alpha[j] - 2.0 * cv * q[j] - 2.0 * A_v * qd[j];
        ks[j] = P.out_sign * G * f1[j];
        vv[j] = P.out_sign * (Gd * f1[j] + G * f2[j] * sj[j]);
    }
    __syncthreads();
    for (int k = lane; k < dim; k += 64) {
        egrad_out[k] = sph_col_dot(P.train + k, ks, (int)n, dim);
        ehess_out[k] = sph_col_dot(P.train + k, vv, (int)n, dim);
    }
}

// preconditioner of manifold_optimize.py:190-193 in ambient coordinates
struct SphPrecon {
    int dim;
    __device__ bool zero_sum(const double* rn) const {
        double s = 0.0;
        for (int e = threadIdx.x; e < dim; e += 64) s += rn[e];
        return wave_sum(s) == 0.0;
    }
    __device__ double entry(double r, bool zs, int) const { return zs ? r + 1e-30 : r; }
};

static __device__ __forceinline__ double dotg(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += 64) s = __builtin_fma(a[e], b[e], s);
    return wave_sum(s);
}

// tCG + proposal + acquisition at the proposal for restart i.  lds: 6 dim doubles; dyn: 3 n doubles.
static __device__ __forceinline__ void sph_propose_body(const double* __restrict__ x, const double* __restrict__ g, double delta_tr,
                                        const double* __restrict__ gc, const double* __restrict__ fc, const SphAcq& P, const SphWs& w,
                                        int64_t i, int64_t R, int C, int neq, double delta_cons, double theta, double kappa,
                                        int mininner, int maxinner, double* lds, double* dyn, int exact_hessian,
                                        bool x_unchanged = false) {
    const int dim = P.dim;
    double* Hd = lds;
    double* dl = Hd + dim;
    double* s0 = dl + dim;
    double* s1 = s0 + dim;
    double* s2 = s1 + dim;
    double* xs = s2 + dim;       // x staged in LDS (read by every lane)
    double* gi = w.g + i * dim;
    double* eta = w.eta + i * dim;
    double* heta = w.heta + i * dim;
    double* rr = w.r + i * dim;
    double* delta = w.delta + i * dim;
    double* sc = w.scal + i * SC_COUNT;
    SphPrecon pc{dim};
    // ---- begin (robust_trust_regions.py:430-470)
    for (int e = threadIdx.x; e < dim; e += 64) {
        xs[e] = x[e];
        s2[e] = g[e];
        gi[e] = g[e];
        rr[e] = g[e];
        eta[e] = 0.0;
        heta[e] = 0.0;
    }
    for (int k = 0; k < C; ++k)
        for (int e = threadIdx.x; e < dim; e += 64) w.gc[((int64_t)k * R + i) * dim + e] = gc[((int64_t)k * R + i) * dim + e];
    __syncthreads();
    const double r2 = dotg(s2, s2, dim);
    const bool zs = pc.zero_sum(s2);
    double zr = 0.0;
    for (int e = threadIdx.x; e < dim; e += 64) {
        const double z = pc.entry(s2[e], zs, e);
        delta[e] = -z;
        zr = __builtin_fma(z, s2[e], zr);
    }
    zr = wave_sum(zr);
    if (threadIdx.x == 0) {
        sc[SC_DELTA] = delta_tr;
        sc[SC_E_PE] = 0.0;
        sc[SC_E_PD] = 0.0;
        sc[SC_D_PD] = zr;
        sc[SC_Z_R] = zr;
        sc[SC_MODEL] = 0.0;
        sc[SC_NORM_R0] = __builtin_sqrt(r2 > 0.0 ? r2 : 0.0);
        w.stop[i] = TCG_MAX_INNER_ITER;
        w.running[i] = 1;
        for (int k = 0; k < C; ++k) { w.fc[i * C + k] = fc[i * C + k]; w.fcg_pe[i * C + k] = 0.0; }
    }
    __syncthreads();
    TcgVecs v{gi, eta, heta, rr, delta, w.gc + i * dim, (int64_t)R * dim, sc, w.fc + i * C, w.fcg_pe + i * C, w.stop + i, w.running + i};
    double* xfd = w.x_fd + i * dim;
    double* egfd = w.eg_fd + i * dim;
    for (int it = 0; it < maxinner; ++it) {
        for (int e = threadIdx.x; e < dim; e += 64) dl[e] = delta[e];
        __syncthreads();
        if (it == 0 && x_unchanged) {
            // after a rejected proposal tCG restarts from eta = 0 with the same x and g (only the radius changed): its first
            // direction and H delta_0 are bit for bit those of the previous iteration - no acquisition evaluation needed
            for (int e = threadIdx.x; e < dim; e += 64) Hd[e] = w.hd0[i * dim + e];
            __syncthreads();
            const bool running_c = tcg_step_core(v, dim, C, Hd, dl, s0, s1, s2, neq, delta_cons, theta, kappa, mininner, it, pc);
            __syncthreads();
            if (!running_c) break;
            continue;
        }
        if (exact_hessian) {
            // Riemannian Hessian ([3P] Sphere.ehess2rhess): proj_x(ehess delta) - <x, egrad> delta
            sph_acq_hess(xs, dl, P, egfd, xfd, dyn);               // egfd <- egrad(x), xfd <- ehess(x) delta
            __syncthreads();
            const double xe = dotg(xs, egfd, dim);
            const double xh = dotg(xs, xfd, dim);
            for (int e = threadIdx.x; e < dim; e += 64) {
                Hd[e] = (xfd[e] - xh * xs[e]) - xe * dl[e];
                if (it == 0) w.hd0[i * dim + e] = Hd[e];
            }
            __syncthreads();
            const bool running_e = tcg_step_core(v, dim, C, Hd, dl, s0, s1, s2, neq, delta_cons, theta, kappa, mininner, it, pc);
            __syncthreads();
            if (!running_e) break;
            continue;
        }
        // FD point (approximate_hessian.py:30-47): c = 2^-14 / |delta|, x1 = retr(x, c delta)
        const double nrm = __builtin_sqrt(dotg(dl, dl, dim));
        const bool tiny = nrm < 1e-15;
        const double c = 6.103515625e-05 / (tiny ? 1.0 : nrm);      // 2^-14 (rounds 2-4 had 2^-13 here)
        double yy = 0.0;
        for (int e = threadIdx.x; e < dim; e += 64) { const double y = xs[e] + c * dl[e]; s0[e] = y; yy = __builtin_fma(y, y, yy); }
        const double inv = 1.0 / __builtin_sqrt(wave_sum(yy));
        for (int e = threadIdx.x; e < dim; e += 64) { s0[e] *= inv; xfd[e] = s0[e]; }
        __syncthreads();
        sph_acq_eval(s0, P, w.val_fd + i, egfd, dyn);
        __syncthreads();
        // Hd = transp(x1 -> x, proj_x1(eg1)) / c - g / c
        const double a1 = dotg(s0, egfd, dim);
        for (int e = threadIdx.x; e < dim; e += 64) s1[e] = egfd[e] - a1 * s0[e];
        __syncthreads();
        const double a0 = dotg(xs, s1, dim);
        for (int e = threadIdx.x; e < dim; e += 64) {
            Hd[e] = tiny ? 0.0 : (s1[e] - a0 * xs[e]) / c - gi[e] / c;
            if (it == 0) w.hd0[i * dim + e] = Hd[e];
        }
        __syncthreads();
        const bool running = tcg_step_core(v, dim, C, Hd, dl, s0, s1, s2, neq, delta_cons, theta, kappa, mininner, it, pc);
        __syncthreads();
        if (!running) break;
    }
    // ---- proposal and model decrease
    const double ge = dotg(gi, eta, dim);
    const double ehe = dotg(eta, heta, dim);
    double yy = 0.0;
    for (int e = threadIdx.x; e < dim; e += 64) { const double y = xs[e] + eta[e]; s0[e] = y; yy = __builtin_fma(y, y, yy); }
    const double inv = 1.0 / __builtin_sqrt(wave_sum(yy));
    double* xp = w.x_prop + i * dim;
    for (int e = threadIdx.x; e < dim; e += 64) { s0[e] *= inv; xp[e] = s0[e]; }
    if (threadIdx.x == 0) w.rhoden[i] = -ge - 0.5 * ehe;
    __syncthreads();
    sph_acq_eval(s0, P, w.fx_prop + i, w.eg_prop + i * dim, dyn);
}

static __device__ __forceinline__ bool sph_update_body(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g, double* __restrict__ ng,
                                       double* __restrict__ delta_tr, int64_t* __restrict__ iters, bool inval, const SphWs& w, int64_t i,
                                       int dim, int C, double delta_bar, double rho_prime, double rho_regularization, double mingradnorm,
                                       int64_t maxiter, bool* accepted = nullptr) {
    const double fx0 = *fx;
    const double fxp = inval ? __builtin_inf() : w.fx_prop[i];
    const double rho_reg = (__builtin_fabs(fx0) > 1.0 ? __builtin_fabs(fx0) : 1.0) * 2.220446049250313e-16 * rho_regularization;
    const double rhonum = (fx0 - fxp) + rho_reg;
    const double rhoden = w.rhoden[i] + rho_reg;
    const bool model_decreased = rhoden >= 0.0;
    const double rho = rhoden == 0.0 ? __builtin_nan("") : rhonum / rhoden;
    const bool shrink = (rho < 0.25) || !model_decreased || (rho != rho) || inval;
    const int stop_inner = w.stop[i];
    const bool boundary = stop_inner == TCG_NEGATIVE_CURVATURE || stop_inner == TCG_EXCEEDED_TR ||
                          (C > 0 && stop_inner == TCG_REACHED_CONSTRAINTS);
    const bool grow = !shrink && rho > 0.75 && boundary;
    const double D0 = *delta_tr;
    const double Dn = shrink ? D0 / 4 : (grow ? (2 * D0 < delta_bar ? 2 * D0 : delta_bar) : D0);
    const bool accept = model_decreased && rho > rho_prime;
    if (accepted) *accepted = accept;                  // (the same value in every lane)
    double ngi = *ng;
    const int64_t it = *iters + 1;
    __syncthreads();
    if (accept) {
        const double* xp = w.x_prop + i * dim;
        const double* eg = w.eg_prop + i * dim;
        const double a = dotg(xp, eg, dim);
        double s = 0.0;
        for (int e = threadIdx.x; e < dim; e += 64) {
            const double gg = eg[e] - a * xp[e];                 // egrad2rgrad = proj
            x[e] = xp[e];
            g[e] = gg;
            s = __builtin_fma(gg, gg, s);
        }
        ngi = __builtin_sqrt(wave_sum(s));
    }
    if (threadIdx.x == 0) {
        *delta_tr = Dn;
        if (accept) { *fx = fxp; *ng = ngi; }
        *iters = it;
    }
    __syncthreads();
    return !(ngi < mingradnorm || it >= maxiter);
}

__global__ __launch_bounds__(64) void sphere_acq_kernel(const double* __restrict__ x, SphAcq P, double* __restrict__ value,
                                                        double* __restrict__ grad, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    double* xs = dyn + 3 * P.n;
    const int64_t i = blockIdx.x;
    for (int e = threadIdx.x; e < P.dim; e += 64) xs[e] = x[i * P.dim + e];
    __syncthreads();
    sph_acq_eval(xs, P, value + i, grad ? grad + i * P.dim : nullptr, dyn);
}

__global__ __launch_bounds__(64) void sphere_tr_propose_kernel(const double* __restrict__ x, const double* __restrict__ g,
                                                               const double* __restrict__ delta_tr, const uint8_t* __restrict__ active,
                                                               const double* __restrict__ gc, const double* __restrict__ fc, SphAcq P,
                                                               void* wsbase, double* __restrict__ x_prop, int64_t R, int C, int neq,
                                                               double delta_cons, double theta, double kappa, int mininner, int maxinner,
                                                               int* __restrict__ any_active, int exact_hessian) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    if (i == 0 && threadIdx.x == 0) *any_active = 0;
    if (active[i] == 0) return;
    SphWs w = sph_layout(wsbase, R, P.dim, C);
    sph_propose_body(x + i * P.dim, g + i * P.dim, delta_tr[i], gc, fc, P, w, i, R, C, neq, delta_cons, theta, kappa, mininner, maxinner,
                     dyn + 7 * P.n, dyn, exact_hessian);
    __syncthreads();
    for (int e = threadIdx.x; e < P.dim; e += 64) x_prop[i * P.dim + e] = w.x_prop[i * P.dim + e];
}

__global__ __launch_bounds__(64) void sphere_tr_update_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                              double* __restrict__ ng, double* __restrict__ delta_tr,
                                                              uint8_t* __restrict__ active, int64_t* __restrict__ iters,
                                                              const uint8_t* __restrict__ invalid, void* wsbase, int64_t R, int dim, int C,
                                                              double delta_bar, double rho_prime, double rho_regularization,
                                                              double mingradnorm, int64_t maxiter, int* __restrict__ any_active) {
    const int64_t i = blockIdx.x;
    if (active[i] == 0) return;
    SphWs w = sph_layout(wsbase, R, dim, C);
    const bool inval = invalid != nullptr && invalid[i] != 0;
    const bool still = sph_update_body(x + i * dim, fx + i, g + i * dim, ng + i, delta_tr + i, iters + i, inval, w, i, dim, C, delta_bar,
                                       rho_prime, rho_regularization, mingradnorm, maxiter);
    if (threadIdx.x == 0) {
        if (!still) active[i] = 0;
        else atomicOr(any_active, 1);
    }
}

// LAT (the latency regime: the symmetric inverse handed over, everything fits, <= 2048 restarts): A and the per-restart workspace live in the block's LDS
// (the solve owns its restart from the first iteration to the last: nothing in the workspace has to survive the launch), known at compile time so that
// their accesses are ds_read / ds_write and not flat loads (see spd_tr_solve_kernel).  Dynamic LDS: 7 n + 6 dim doubles, then n^2 + 2 n dim (the training points in both
// layouts) + the workspace of ONE restart.
static inline size_t sph_solve_lds(int64_t n, int dim, int64_t r, bool sym, bool* lat) {
    const size_t base = (size_t)(7 * n + 6 * dim) * sizeof(double);
    const size_t extra = (size_t)(n * n + 2 * n * dim) * sizeof(double) + ((sph_layout(nullptr, 1, dim, 0).bytes + 15) & ~(size_t)15);
    *lat = sym && r <= 2048 && base + extra <= 56 * 1024;
    return base + (*lat ? extra : 0);
}

template <bool LAT>
__global__ __launch_bounds__(64) void sphere_tr_solve_kernel(double* __restrict__ x, double* __restrict__ fx, double* __restrict__ g,
                                                             double* __restrict__ ng, double* __restrict__ delta_tr,
                                                             uint8_t* __restrict__ active, int64_t* __restrict__ iters, SphAcq P,
                                                             void* wsbase, int64_t R, double theta, double kappa, int mininner,
                                                             int maxinner, double delta_bar, double rho_prime, double rho_regularization,
                                                             double mingradnorm, int64_t maxiter, int exact_hessian,
                                                             double* __restrict__ rec, int64_t rec_cap) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    if (active[i] == 0) return;
    SphAcq Ps = P;
    SphWs w;
    int64_t iw = i, Rw = R;
    if constexpr (LAT) {
        double* gl = dyn + 7 * P.n + 6 * P.dim;
        const int64_t nn = P.n * P.n;
        for (int64_t e = threadIdx.x; e < nn; e += 64) gl[e] = P.linv[e];
        Ps.linv = gl;
        Ps.linv_t = gl;
        double* tr = gl + nn;                    // the training points in both layouts (n x dim, dim x n)
        double* trt = tr + P.n * P.dim;
        for (int64_t e = threadIdx.x; e < P.n * P.dim; e += 64) {
            tr[e] = P.train[e];
            trt[e] = P.train_t[e];
        }
        Ps.train = tr;
        Ps.train_t = trt;
        double* wb = trt + P.n * P.dim;
        w = sph_layout(wb, 1, P.dim, 0);
        for (size_t e = threadIdx.x; e < (w.bytes + 7) / 8; e += 64) wb[e] = 0.0;
        iw = 0;
        Rw = 1;
        __syncthreads();
    } else {
        w = sph_layout(wsbase, R, P.dim, 0);
    }
    bool x_unchanged = false;         // wave-uniform: the previous proposal of this launch was rejected
    int64_t rec_k = rec != nullptr ? iters[i] : 0;      // gabo_tr_solve_record: index of the outer iteration being recorded
    for (;;) {
        sph_propose_body(x + i * P.dim, g + i * P.dim, delta_tr[i], nullptr, nullptr, Ps, w, iw, Rw, 0, 0, 1e-6, theta, kappa, mininner,
                         maxinner, dyn + 7 * P.n, dyn, exact_hessian, x_unchanged);
        __syncthreads();
        if (rec != nullptr && rec_k < rec_cap) {          // (the iterate, its radius and the stop reason of the tCG run that made the proposal)
            double* rr = rec + (rec_k * R + i) * (P.dim + 2);
            for (int e = threadIdx.x; e < P.dim; e += 64) rr[e] = x[i * P.dim + e];
            if (threadIdx.x == 0) {
                rr[P.dim] = delta_tr[i];
                rr[P.dim + 1] = (double)w.stop[iw];
            }
        }
        ++rec_k;
        bool accepted = false;
        const bool still = sph_update_body(x + i * P.dim, fx + i, g + i * P.dim, ng + i, delta_tr + i, iters + i, false, w, iw, P.dim, 0,
                                           delta_bar, rho_prime, rho_regularization, mingradnorm, maxiter, &accepted);
        if (!still) break;
        x_unchanged = !accepted;
    }
    if (threadIdx.x == 0) active[i] = 0;
}

void tr_record_take(double** buffer, int64_t* capacity);      // spd_tr.hip

static int sph_acq_ok(const SphAcq* a) {
    if (!a || a->n < 1 || a->n > 4096 || a->dim < 2 || a->dim > 512 || !a->train || !a->train_t || !a->alpha) return GABO_ERR_ARG;
    const int out = a->flags & GABO_OUT_MASK;
    if ((a->flags & ~GABO_OUT_MASK) || (out != GABO_OUT_GAUSSIAN && out != GABO_OUT_LAPLACE)) return GABO_ERR_ARG;
    if (a->kind != GABO_ACQ_EXPECTED_IMPROVEMENT && a->kind != GABO_ACQ_POSTERIOR_MEAN) return GABO_ERR_ARG;
    if (a->kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!a->linv || !a->linv_t)) return GABO_ERR_ARG;
    if ((size_t)(7 * a->n + 6 * a->dim) * sizeof(double) > 150 * 1024) return GABO_ERR_ARG;
    return GABO_OK;
}

}  // namespace gabo

extern "C" {

int gabo_sphere_acq_eval(const double* x, const gabo_sphere_acq_params* acq, double* value, double* grad, int64_t r,
                         gabo_stream_t stream) {
    int rc = gabo::sph_acq_ok(acq);
    if (rc != GABO_OK) return rc;
    if (r < 0 || r > 0x7fffffffLL) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !value) return GABO_ERR_ARG;
    size_t lds = (size_t)(3 * acq->n + acq->dim) * sizeof(double);
    hipLaunchKernelGGL(gabo::sphere_acq_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, x, *acq, value, grad, r);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

size_t gabo_sphere_tr_workspace_bytes(int64_t r, int dim, int n_constraints) {
    if (r < 0 || dim < 1 || n_constraints < 0) return 0;
    return gabo::sph_layout(nullptr, r, dim, n_constraints).bytes;
}

size_t gabo_sphere_tr_stop_offset(int64_t r, int dim, int n_constraints) {
    if (r < 0 || dim < 1 || n_constraints < 0) return 0;
    return (size_t)((char*)gabo::sph_layout(nullptr, r, dim, n_constraints).stop - (char*)nullptr);
}

int gabo_sphere_tr_propose(const double* x, const double* grad, const double* trust_radius, const uint8_t* active,
                           const double* cons_grads, const double* cons_values, const gabo_sphere_acq_params* acq, void* workspace,
                           size_t workspace_bytes, double* x_prop, int64_t r, int n_constraints, int n_equalities, double delta_cons,
                           double theta, double kappa, int mininner, int maxinner, int exact_hessian, int* any_active, gabo_stream_t stream) {
    int rc = gabo::sph_acq_ok(acq);
    if (rc != GABO_OK) return rc;
    if (r < 0 || r > 0x7fffffffLL || n_constraints < 0 || n_constraints > gabo::kMaxCons || n_equalities < 0 ||
        n_equalities > n_constraints || maxinner < 1)
        return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !grad || !trust_radius || !active || !workspace || !x_prop || !any_active ||
        (n_constraints > 0 && (!cons_grads || !cons_values)))
        return GABO_ERR_ARG;
    if (workspace_bytes < gabo_sphere_tr_workspace_bytes(r, acq->dim, n_constraints)) return GABO_ERR_ARG;
    size_t lds = (size_t)(7 * acq->n + 6 * acq->dim) * sizeof(double);
    hipLaunchKernelGGL(gabo::sphere_tr_propose_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, x, grad, trust_radius, active,
                       cons_grads, cons_values, *acq, workspace, x_prop, r, n_constraints, n_equalities, delta_cons, theta, kappa,
                       mininner, maxinner, any_active, exact_hessian);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_sphere_tr_update(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                          const uint8_t* invalid, void* workspace, int64_t r, int dim, int n_constraints, double delta_bar,
                          double rho_prime, double rho_regularization, double mingradnorm, int64_t maxiter, int* any_active,
                          gabo_stream_t stream) {
    if (dim < 2 || dim > 512) return GABO_ERR_DIM;
    if (r < 0 || r > 0x7fffffffLL || n_constraints < 0 || n_constraints > gabo::kMaxCons) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !fx || !grad || !grad_norm || !trust_radius || !active || !iters || !workspace || !any_active) return GABO_ERR_ARG;
    hipLaunchKernelGGL(gabo::sphere_tr_update_kernel, dim3((unsigned)r), dim3(64), 0, (hipStream_t)stream, x, fx, grad, grad_norm,
                       trust_radius, active, iters, invalid, workspace, r, dim, n_constraints, delta_bar, rho_prime, rho_regularization,
                       mingradnorm, maxiter, any_active);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_sphere_tr_solve(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                         const gabo_sphere_acq_params* acq, void* workspace, size_t workspace_bytes, int64_t r, double theta, double kappa,
                         int mininner, int maxinner, int exact_hessian, double delta_bar, double rho_prime, double rho_regularization,
                         double mingradnorm, int64_t maxiter, gabo_stream_t stream) {
    double* rec = nullptr;
    int64_t rec_cap = 0;
    gabo::tr_record_take(&rec, &rec_cap);                 // gabo_tr_solve_record (spd_tr.hip): consumed by this call whether it launches or not
    int rc = gabo::sph_acq_ok(acq);
    if (rc != GABO_OK) return rc;
    if (r < 0 || r > 0x7fffffffLL || maxinner < 1 || maxiter < 1) return GABO_ERR_ARG;
    if (r == 0) return GABO_OK;
    if (!x || !fx || !grad || !grad_norm || !trust_radius || !active || !iters || !workspace) return GABO_ERR_ARG;
    if (workspace_bytes < gabo_sphere_tr_workspace_bytes(r, acq->dim, 0)) return GABO_ERR_ARG;
    bool lat = false;
    const size_t lds = gabo::sph_solve_lds(acq->n, acq->dim, r, acq->linv != nullptr && acq->linv == acq->linv_t, &lat);
    if (lat)
        hipLaunchKernelGGL(gabo::sphere_tr_solve_kernel<true>, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, x, fx, grad, grad_norm,
                           trust_radius, active, iters, *acq, workspace, r, theta, kappa, mininner, maxinner, delta_bar, rho_prime,
                           rho_regularization, mingradnorm, maxiter, exact_hessian, rec, rec_cap);
    else
        hipLaunchKernelGGL(gabo::sphere_tr_solve_kernel<false>, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, x, fx, grad, grad_norm,
                           trust_radius, active, iters, *acq, workspace, r, theta, kappa, mininner, maxinner, delta_bar, rho_prime,
                           rho_regularization, mingradnorm, maxiter, exact_hessian, rec, rec_cap);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // extern "C"
