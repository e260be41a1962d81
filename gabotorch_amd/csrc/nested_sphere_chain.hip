// Nested-sphere mappings over ALL their levels in one launch (HD-GaBO on the sphere: examples/hd_bo_sphere/benchmark_examples/hd_gabo_sphere.py).
//   projection_from_sphere_to_subsphere     nested_mappings/nested_spheres_utils.py:117-146 (levels :68-114)
//   projection_from_subsphere_to_sphere     nested_mappings/nested_spheres_utils.py:182-218 (levels :149-179)
//   rotation_from_sphere_points_torch       Riemannian_utils/sphere_utils_torch.py:58-93
//   min_error_reconstruction_cost           nested_mappings/nested_spheres_optimization.py:20-38
// The reference walks the D - latent levels in Python, a dozen torch operations each, and differentiates them by autograd; with the
// per-level HIP epilogue of nested_sphere.hip that was still ~16 launches per level and direction: 0.2 / 1.2 / 2.7 s per surrogate fit and
// 0.04 / 0.4 / 4.2 s per reconstruction optimisation at D = 5 / 21 / 51.  A level costs O(d) flops per point: here one wave owns one
// point, walks all the levels with the intermediate points in LDS (sum_k d_k <= D^2 / 2 doubles), and walks them back for the gradient.
//
// The rotation R(axis -> north pole) acts in the plane span{axis, north} only (Jung, Dryden & Marron 2012): with t = clamp(axis_d),
// e = (axis - t north) / |axis - t north|, s = sqrt(1 - t^2):  R p = p + (s <p,e> + (t-1) p_d) north + (-s p_d + (t-1) <p,e>) e,
// and R^T is the same with s -> -s.  The `frame` of a level is (e, s, t); frames of all levels are packed like the axes (level k has
// dimension D - k, offset sum_{j<k} (D - j)), followed by the (s, t) pairs.
#include <cstring>

#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

constexpr double kChainEps = 1e-6;                // the reference's "+ 1e-6 * ones" (nested_spheres_utils.py:56,105,110)
constexpr double kChainClamp = 1.0 - 1e-15;       // sphere_utils_torch.py:53,78-80

__host__ __device__ inline int chain_offset(int D, int k) { return k * D - k * (k - 1) / 2; }      // sum_{j<k} (D - j)

static __device__ __forceinline__ double wave_sum(double v) { return wave_allsum(v); }      // DPP row operations + two v_readlane (gabo_device.hpp)

// frames of the L levels from their axes: one wave per level
__global__ __launch_bounds__(64) void nested_sphere_frames_kernel(const double* __restrict__ axes, double* __restrict__ frames, int D, int L) {
    const int k = blockIdx.x, d = D - k, off = chain_offset(D, k);
    const double* a = axes + off;
    double* e = frames + off;
    double* st = frames + chain_offset(D, L) + 2 * k;
    const double raw = a[d - 1];
    const double t = raw > kChainClamp ? kChainClamp : (raw < -kChainClamp ? -kChainClamp : raw);
    double nn = 0.0;
    for (int i = threadIdx.x; i < d; i += 64) {
        const double v = a[i] - (i == d - 1 ? t : 0.0);
        nn = __builtin_fma(v, v, nn);
    }
    nn = __builtin_sqrt(wave_sum(nn));
    for (int i = threadIdx.x; i < d; i += 64) e[i] = (a[i] - (i == d - 1 ? t : 0.0)) / nn;
    if (threadIdx.x == 0) { st[0] = __builtin_sqrt((1.0 - t) * (1.0 + t)); st[1] = t; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Reconstruction: z (N x (D-L)) lifted level by level to S^(D-1), cost_p = sum_n acos(clamp(<x_n, rec_n>))^2 and d cost_p / d r_pk.
// Block (64 threads) per (parameter set p, data point n); records[(p N + n) (1 + L)] = (cost_n, g_r[0..L)); the last block of a
// parameter set adds them in data order.
__global__ __launch_bounds__(64) void nested_sphere_reconstruction_kernel(const double* __restrict__ xdata, const double* __restrict__ z,
                                                                         const double* __restrict__ frames, const double* __restrict__ dists,
                                                                         double* __restrict__ cost, double* __restrict__ grad, int P, int N, int D,
                                                                         int L, int* __restrict__ counters, double* __restrict__ records) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // level inputs: the input of level k (dimension D - k - 1) at in_off(k) = sum_{j>k} (D - j - 1); then the running point and its adjoint
    const int lat = D - L;
    double* xin = lds;
    const int total_in = L * lat + L * (L - 1) / 2;           // sum_{k=0}^{L-1} (D - k - 1)
    double* cur = xin + total_in;                             // D
    double* g = cur + D;                                      // D
    double* srt = g + D;                                      // L : sin r_k, then L : cos r_k (one level per lane, once)
    double* crt = srt + L;
    double* fr = crt + L;                                     // the frames: every level reads its e from LDS, not from L2 (48 dependent round trips)
    const int p = blockIdx.x / N, n = blockIdx.x - p * N;
    const int lane = threadIdx.x;
    const int nframes = chain_offset(D, L) + 2 * L;
    for (int i = lane; i < nframes; i += 64) fr[i] = frames[i];
    const double* stv = fr + chain_offset(D, L);
    const bool want_grad = grad != nullptr;
    for (int i = lane; i < lat; i += 64) cur[i] = z[(size_t)n * lat + i];
    for (int k = lane; k < L; k += 64) {
        const double r = dists[(size_t)p * L + k];
        srt[k] = sin(r);
        crt[k] = cos(r);
    }
    __syncthreads();
    int in_off = 0;
    for (int k = L - 1; k >= 0; --k) {                        // level k: S^(d-2) -> S^(d-1), d = D - k
        const int d = D - k;
        const double* e = fr + chain_offset(D, k);
        const double s = stv[2 * k], t = stv[2 * k + 1];
        const double sr = srt[k], cr = crt[k];
        double dot = 0.0;
        for (int i = lane; i < d - 1; i += 64) {
            const double xi = cur[i];
            xin[in_off + i] = xi;
            dot = __builtin_fma(xi, e[i], dot);
        }
        dot = wave_sum(dot);
        const double el = e[d - 1];
        const double a_ = cr, b = sr * dot + cr * el;
        const double cy = -s * b + (t - 1.0) * a_, ce = s * a_ + (t - 1.0) * b;        // R^T: s -> -s
        __syncthreads();
        for (int i = lane; i < d - 1; i += 64) cur[i] = sr * cur[i] + ce * e[i];
        if (lane == 0) cur[d - 1] = cr + cy + ce * el;
        __syncthreads();
        in_off += d - 1;
    }
    // distance to the data point
    double c = 0.0;
    for (int i = lane; i < D; i += 64) c = __builtin_fma(xdata[(size_t)n * D + i], cur[i], c);
    c = wave_sum(c);
    const bool inside = c >= -kChainClamp && c <= kChainClamp;
    const double cc = c < -kChainClamp ? -kChainClamp : (c > kChainClamp ? kChainClamp : c);
    const double theta = acos(cc);
    double* R = records + ((size_t)p * N + n) * (1 + L);
    if (lane == 0) R[0] = theta * theta;
    if (want_grad) {
        const double gc = inside ? -2.0 * theta / __builtin_sqrt((1.0 - cc) * (1.0 + cc)) : 0.0;
        for (int i = lane; i < D; i += 64) g[i] = gc * xdata[(size_t)n * D + i];
        __syncthreads();
        for (int k = 0; k < L; ++k) {                         // back through the levels, last applied first
            const int d = D - k;
            in_off -= d - 1;
            const double* e = fr + chain_offset(D, k);
            const double s = stv[2 * k], t = stv[2 * k + 1];
            const double sr = srt[k], cr = crt[k];
            double ge = 0.0;
            for (int i = lane; i < d; i += 64) ge = __builtin_fma(g[i], e[i], ge);
            ge = wave_sum(ge);                                 // g_ce
            const double gcy = g[d - 1];
            const double gb = -s * gcy + (t - 1.0) * ge, ga = (t - 1.0) * gcy + s * ge;
            // g_v = g + g_b e (+ g_a on the last coordinate);  v = [sin r x, cos r]
            double gx_dot = 0.0;
            __syncthreads();
            for (int i = lane; i < d - 1; i += 64) {
                const double gv = __builtin_fma(gb, e[i], g[i]);
                gx_dot = __builtin_fma(gv, xin[in_off + i], gx_dot);
                g[i] = sr * gv;
            }
            gx_dot = wave_sum(gx_dot);
            const double gv_last = gcy + gb * e[d - 1] + ga;
            if (lane == 0) R[1 + k] = cr * gx_dot - sr * gv_last;
            __syncthreads();
        }
    }
    // the last block of this parameter set adds the records in data order
    __threadfence();
    __shared__ int last;
    if (lane == 0) last = atomicAdd(counters + p, 1) == N - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (lane == 0) counters[p] = 0;
    const double* Rp = records + (size_t)p * N * (1 + L);
    for (int q = lane; q < (want_grad ? 1 + L : 1); q += 64) {
        double sum = 0.0;
        for (int m = 0; m < N; ++m) sum += Rp[(size_t)m * (1 + L) + q];
        if (q == 0) cost[p] = sum;
        else grad[(size_t)p * L + q - 1] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Projection S^(D-1) -> S^(D-L-1) through all levels (the per-point half of NestedSphereGaussianKernel.forward,
// kernel_utils/kernels_nested_sphere.py:125-152), one wave per point.  Level k (d = D - k), input p in S^(d-1):
//   U = R(axis_k -> north) p;  theta = acos(clamp(U_d));  w = sin r / ((sin theta + 1e-6)(sin r + 1e-6)) U[0:d-1];  z = w / (|w| + 1e-6)
// (the mode-0 epilogue of nested_sphere.hip behind the rank-2 rotation).  store != NULL keeps the level inputs (n x sum_k d_k, packed
// like the axes) for the backward pass.
struct ChainLevel {
    double a_, b, cy, ce, tc, st, sr, A;
    bool inside;
};

// (sin theta of theta = acos(tc) is sqrt(1 - tc^2): no inverse trigonometric function on the 48-level critical path; sr = sin r_k is the
// same for every point and comes from a table the wave fills once, one level per lane)
static __device__ __forceinline__ ChainLevel chain_level_forward(const double* p, const double* e, double s, double t, double sr, int d, int lane) {
    ChainLevel lv;
    double dot = 0.0;
    for (int i = lane; i < d; i += 64) dot = __builtin_fma(p[i], e[i], dot);
    lv.b = wave_sum(dot);
    lv.a_ = p[d - 1];
    lv.cy = s * lv.b + (t - 1.0) * lv.a_;
    lv.ce = -s * lv.a_ + (t - 1.0) * lv.b;
    const double tl = lv.a_ + lv.cy + lv.ce * e[d - 1];
    lv.inside = tl < kChainClamp && tl > -kChainClamp;
    lv.tc = tl > kChainClamp ? kChainClamp : (tl < -kChainClamp ? -kChainClamp : tl);
    lv.st = sqrt_nz((1.0 - lv.tc) * (1.0 + lv.tc));             // (> 0 inside the clamp; seed + Goldschmidt: ~1 ulp, a third of the IEEE sequence)
    lv.sr = sr;
    lv.A = lv.sr * rcp((lv.st + kChainEps) * (lv.sr + kChainEps));
    return lv;
}

__global__ __launch_bounds__(64) void nested_sphere_project_kernel(const double* __restrict__ x, const double* __restrict__ frames,
                                                                  const double* __restrict__ dists, double* __restrict__ z,
                                                                  double* __restrict__ store, int D, int L) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* cur = lds;                // D
    double* srt = cur + D;            // L : sin r_k
    // (the frames are read from L2 here: staging their 10 KB in LDS per block pays for the few blocks of a reconstruction launch, not for
    // thousands of points - measured 74 -> 98 us for 4096 points at D = 51)
    const double* fr = frames;
    const int n = blockIdx.x, lane = threadIdx.x;
    const int total = chain_offset(D, L);
    const double* stv = fr + total;
    for (int i = lane; i < D; i += 64) cur[i] = x[(size_t)n * D + i];
    for (int k = lane; k < L; k += 64) srt[k] = sin(dists[k]);
    __syncthreads();
    for (int k = 0; k < L; ++k) {
        const int d = D - k;
        const double* e = fr + chain_offset(D, k);
        if (store) for (int i = lane; i < d; i += 64) store[(size_t)n * total + chain_offset(D, k) + i] = cur[i];
        const ChainLevel lv = chain_level_forward(cur, e, stv[2 * k], stv[2 * k + 1], srt[k], d, lane);
        double nn = 0.0;
        __syncthreads();
        for (int i = lane; i < d - 1; i += 64) {
            const double w = lv.A * __builtin_fma(lv.ce, e[i], cur[i]);
            cur[i] = w;
            nn = __builtin_fma(w, w, nn);
        }
        const double inv = rcp(sqrt_pos(wave_sum(nn)) + kChainEps);
        for (int i = lane; i < d - 1; i += 64) cur[i] *= inv;
        __syncthreads();
    }
    for (int i = lane; i < D - L; i += 64) z[(size_t)n * (D - L) + i] = cur[i];
}

// Lift S^(D-L-1) -> S^(D-1) through all levels (projection_from_subsphere_to_sphere, nested_spheres_utils.py:182-218), one wave per point:
// level k (applied from k = L-1 down to 0) maps x in S^(d-2) to R(north -> axis_k) [sin r_k x, cos r_k] in S^(d-1), d = D - k.
// levels_out != NULL keeps every level's output (packed like the axes: level 0's, the final point, first).
__global__ __launch_bounds__(64) void nested_sphere_lift_kernel(const double* __restrict__ z, const double* __restrict__ frames,
                                                               const double* __restrict__ dists, double* __restrict__ xout,
                                                               double* __restrict__ levels_out, int D, int L) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* cur = lds;                // D
    double* srt = cur + D;            // L, L : sin r_k, cos r_k
    double* crt = srt + L;
    const double* fr = frames;
    const int n = blockIdx.x, lane = threadIdx.x, lat = D - L;
    const int total = chain_offset(D, L);
    const double* stv = fr + total;
    for (int i = lane; i < lat; i += 64) cur[i] = z[(size_t)n * lat + i];
    for (int k = lane; k < L; k += 64) { srt[k] = sin(dists[k]); crt[k] = cos(dists[k]); }
    __syncthreads();
    for (int k = L - 1; k >= 0; --k) {
        const int d = D - k;
        const double* e = fr + chain_offset(D, k);
        const double s = stv[2 * k], t = stv[2 * k + 1];
        const double sr = srt[k], cr = crt[k];
        double dot = 0.0;
        for (int i = lane; i < d - 1; i += 64) dot = __builtin_fma(cur[i], e[i], dot);
        dot = wave_sum(dot);
        const double el = e[d - 1];
        const double b = sr * dot + cr * el;
        const double cy = -s * b + (t - 1.0) * cr, ce = s * cr + (t - 1.0) * b;
        __syncthreads();
        for (int i = lane; i < d - 1; i += 64) cur[i] = sr * cur[i] + ce * e[i];
        if (lane == 0) cur[d - 1] = cr + cy + ce * el;
        __syncthreads();
        if (levels_out) for (int i = lane; i < d; i += 64) levels_out[(size_t)n * total + chain_offset(D, k) + i] = cur[i];
    }
    if (xout) for (int i = lane; i < D; i += 64) xout[(size_t)n * D + i] = cur[i];
}

// Adjoint of the projection with respect to the frames: gz (n x (D-L)) -> per-point partials [g_e (packed like the axes) | (g_s, g_t) per level]
__global__ __launch_bounds__(64) void nested_sphere_project_backward_kernel(const double* __restrict__ store, const double* __restrict__ frames,
                                                                           const double* __restrict__ dists, const double* __restrict__ gz,
                                                                           double* __restrict__ partial, int D, int L) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* G = lds;                  // D : gradient with respect to the current level's output, then input
    double* U = G + D;                // D
    double* srt = U + D;              // L : sin r_k
    double* fr = srt + L;             // the frames
    const int n = blockIdx.x, lane = threadIdx.x;
    const int total = chain_offset(D, L);
    for (int i = lane; i < total + 2 * L; i += 64) fr[i] = frames[i];
    const double* stv = fr + total;
    double* out = partial + (size_t)n * (total + 2 * L);
    for (int i = lane; i < D - L; i += 64) G[i] = gz[(size_t)n * (D - L) + i];
    for (int k = lane; k < L; k += 64) srt[k] = sin(dists[k]);
    __syncthreads();
    for (int k = L - 1; k >= 0; --k) {
        const int d = D - k;
        const double* e = fr + chain_offset(D, k);
        const double* p = store + (size_t)n * total + chain_offset(D, k);
        const double s = stv[2 * k], t = stv[2 * k + 1];
        const ChainLevel lv = chain_level_forward(p, e, s, t, srt[k], d, lane);
        // recompute U[0:d-1], w = A U, |w|
        double nn = 0.0, gw_dot = 0.0;
        for (int i = lane; i < d - 1; i += 64) {
            const double u = __builtin_fma(lv.ce, e[i], p[i]);
            U[i] = u;
            const double w = lv.A * u;
            nn = __builtin_fma(w, w, nn);
            gw_dot = __builtin_fma(G[i], w, gw_dot);
        }
        nn = wave_sum(nn);
        gw_dot = wave_sum(gw_dot);
        const double nrm = sqrt_pos(nn), den = nrm + kChainEps, iden = rcp(den);
        const double coef = nrm > 0.0 ? gw_dot * rcp(nrm * den * den) : 0.0;
        // z = w / den:  g_w = g_z / den - (g_z . w) w / (|w| den^2);  g_U = A g_w;  g_A = sum g_w U
        double gA = 0.0, gce = 0.0;
        __syncthreads();
        for (int i = lane; i < d - 1; i += 64) {
            const double gw = G[i] * iden - coef * (lv.A * U[i]);
            gA = __builtin_fma(gw, U[i], gA);
            const double gu = lv.A * gw;
            G[i] = gu;                                        // g_U
            gce = __builtin_fma(gu, e[i], gce);
        }
        gA = wave_sum(gA);
        gce = wave_sum(gce);
        const double gu_last = lv.inside ? gA * lv.A * lv.tc * rcp((lv.st + kChainEps) * lv.st) : 0.0;      // cos theta = tc, dtheta/dt = -1 / sin theta
        gce += gu_last * e[d - 1];
        const double gcy = gu_last;
        const double gs = gcy * lv.b - gce * lv.a_, gt = gcy * lv.a_ + gce * lv.b;
        const double gb = gcy * s + gce * (t - 1.0), ga = gcy * (t - 1.0) - gce * s;
        __syncthreads();
        // g_e = ce g_U + g_b p ;  g_p = g_U + g_b e (+ g_a on the last coordinate)
        for (int i = lane; i < d; i += 64) {
            const double gu = i < d - 1 ? G[i] : gu_last;
            out[chain_offset(D, k) + i] = __builtin_fma(lv.ce, gu, gb * p[i]);
            G[i] = __builtin_fma(gb, e[i], gu) + (i == d - 1 ? ga : 0.0);
        }
        if (lane == 0) { out[total + 2 * k] = gs; out[total + 2 * k + 1] = gt; }
        __syncthreads();
    }
}

// Sum of the per-point partials and the adjoint of axis -> frame: one wave per level.  g_axes packed like the axes.
__global__ __launch_bounds__(64) void nested_sphere_axes_adjoint_kernel(const double* __restrict__ axes, const double* __restrict__ frames,
                                                                       const double* __restrict__ partial, double* __restrict__ g_axes, int n,
                                                                       int D, int L) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* ge = lds;                 // d
    const int k = blockIdx.x, d = D - k, off = chain_offset(D, k), lane = threadIdx.x;
    const int total = chain_offset(D, L), rec = total + 2 * L;
    const double* a = axes + off;
    const double* e = frames + off;
    double dot = 0.0;
    for (int i = lane; i < d; i += 64) {
        double s = 0.0;
        for (int q = 0; q < n; ++q) s += partial[(size_t)q * rec + off + i];
        ge[i] = s;
        dot = __builtin_fma(s, e[i], dot);
    }
    dot = wave_sum(dot);
    double gs = 0.0, gt = 0.0;
    for (int q = 0; q < n; ++q) { gs += partial[(size_t)q * rec + total + 2 * k]; gt += partial[(size_t)q * rec + total + 2 * k + 1]; }
    const double raw = a[d - 1];
    const bool inside = raw < kChainClamp && raw > -kChainClamp;
    const double t = frames[total + 2 * k + 1], s = frames[total + 2 * k];
    double nn = 0.0;
    for (int i = lane; i < d; i += 64) {
        const double v = a[i] - (i == d - 1 ? t : 0.0);
        nn = __builtin_fma(v, v, nn);
    }
    nn = __builtin_sqrt(wave_sum(nn));
    __syncthreads();
    for (int i = lane; i < d; i += 64) {
        const double gabar = (ge[i] - dot * e[i]) / nn;      // adjoint of the normalisation
        if (i < d - 1) g_axes[off + i] = gabar;
        else g_axes[off + i] = inside ? gt - gs * t / s : gabar;       // t = clamp(axis_d), s = sqrt(1 - t^2); (axis - t north)_d is constant inside the clamp
    }
}

// d ll / d z_i = sum_j gs_ij f'(<z_i, z_j>) z_j for the Gaussian sphere kernel f(c) = exp(-beta acos(clamp c)^2)  (kernels_sphere.py:71-94),
// gs the symmetrised adjoint of the Gram matrix.  One wave per row i.
__global__ __launch_bounds__(64) void sphere_gram_adjoint_kernel(const double* __restrict__ z, const double* __restrict__ gs, double* __restrict__ gz,
                                                                int n, int lat, double beta) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* zi = lds;                 // lat
    double* acc = zi + lat;           // 64 x lat partial sums, reduced at the end
    const int i = blockIdx.x, lane = threadIdx.x;
    for (int l = lane; l < lat; l += 64) zi[l] = z[(size_t)i * lat + l];
    for (int l = 0; l < lat; ++l) acc[lane * lat + l] = 0.0;
    __syncthreads();
    for (int j = lane; j < n; j += 64) {
        const double* zj = z + (size_t)j * lat;
        double c = 0.0;
        for (int l = 0; l < lat; ++l) c = __builtin_fma(zi[l], zj[l], c);
        if (c > kChainClamp || c < -kChainClamp) continue;   // clamp active: zero derivative (autograd semantics; the diagonal)
        const double th = acos(c);
        const double t1 = -1.0 / __builtin_sqrt((1.0 - c) * (1.0 + c));
        const double w = gs[(size_t)i * n + j] * exp(-((th * th) * beta)) * (-2.0 * beta * th * t1);
        for (int l = 0; l < lat; ++l) acc[lane * lat + l] = __builtin_fma(w, zj[l], acc[lane * lat + l]);
    }
    __syncthreads();
    for (int l = lane; l < lat; l += 64) {
        double s = 0.0;
        for (int q = 0; q < 64; ++q) s += acc[q * lat + l];
        gz[(size_t)i * lat + l] = s;
    }
}

// dynamic LDS of a chain kernel beyond the 64 KB default needs the attribute (D >~ 90)
template <typename K>
static bool chain_lds_ok(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return false;
    return bytes <= 64 * 1024 ||
           hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

int fit_gram_adjoint_launch(const double* kb, const double* wm, double* gs, double* out, double* partial, int* counter, int64_t n, double half_os,
                            double theta, unsigned blocks, hipStream_t s);      // nested_spd_fit.hip

namespace {

size_t chain_align256(size_t b) { return (b + 255) & ~(size_t)255; }

// device workspace of gabo_nested_sphere_fit_evaluate
struct SphereFitLayout {
    size_t axes, frames, z, store, kb, wm, gs, gz, out, partial, counters, mll, total;
    unsigned blocks_gram;
    SphereFitLayout(int64_t n, int D, int L) {
        const size_t tot = (size_t)chain_offset(D, L), nn = (size_t)n * n, lat = (size_t)(D - L);
        blocks_gram = (unsigned)((nn + 255) / 256 > 256 ? 256 : (nn + 255) / 256);
        size_t o = 0;
        auto take = [&o](size_t doubles) { const size_t at = o; o += chain_align256(doubles * sizeof(double)); return at; };
        axes = take(tot + L);
        frames = take(tot + 2 * (size_t)L);
        z = take((size_t)n * lat);
        store = take((size_t)n * tot);
        kb = take(nn);
        wm = take(nn);
        gs = take(nn);
        gz = take((size_t)n * lat);
        out = take(7 + tot);
        const size_t pg = blocks_gram, pp = (size_t)n * (tot + 2 * (size_t)L);
        partial = take(pg > pp ? pg : pp);
        counters = take(2);
        mll = o;
        o += chain_align256(n > GABO_GP_MLL_MAX_N ? gabo_gp_mll_large_workspace_bytes(n) : 0);
        total = o;
    }
};

}  // namespace
}  // namespace gabo

extern "C" {

int gabo_nested_sphere_frames(const double* axes, double* frames, int D, int levels, gabo_stream_t stream) {
    if (D < 2 || D > 4096 || levels < 1 || levels > D - 1) return GABO_ERR_DIM;
    if (!axes || !frames) return GABO_ERR_ARG;
    hipLaunchKernelGGL(gabo::nested_sphere_frames_kernel, dim3((unsigned)levels), dim3(64), 0, (hipStream_t)stream, axes, frames, D, levels);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_nested_sphere_project(const double* x, const double* frames, const double* distances, double* z, double* levels_in, int64_t n, int D,
                               int levels, gabo_stream_t stream) {
    if (D < 2 || D > 4096 || levels < 1 || levels > D - 1) return GABO_ERR_DIM;
    if (n < 0) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    if (!x || !frames || !distances || !z || n > 0x7fffffffLL) return GABO_ERR_ARG;
    const size_t lds = (size_t)(D + levels) * sizeof(double);
    hipLaunchKernelGGL(gabo::nested_sphere_project_kernel, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x, frames,
                       distances, z, levels_in, D, levels);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_nested_sphere_lift(const double* x_subsphere, const double* frames, const double* distances, double* x, double* levels_out, int64_t n,
                            int D, int levels, gabo_stream_t stream) {
    if (D < 2 || D > 4096 || levels < 1 || levels > D - 1) return GABO_ERR_DIM;
    if (n < 0) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    if (!x_subsphere || !frames || !distances || (!x && !levels_out) || n > 0x7fffffffLL) return GABO_ERR_ARG;
    const size_t lds = (size_t)(D + 2 * levels) * sizeof(double);
    hipLaunchKernelGGL(gabo::nested_sphere_lift_kernel, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, x_subsphere,
                       frames, distances, x, levels_out, D, levels);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

size_t gabo_nested_sphere_reconstruction_workspace_bytes(int64_t P, int64_t N, int D, int levels) {
    if (P <= 0 || N <= 0 || D < 2 || levels < 1 || levels > D - 1) return 0;
    const size_t counters = ((size_t)P * sizeof(int) + 15) / 16 * 16;
    return counters + (size_t)P * (size_t)N * (size_t)(1 + levels) * sizeof(double);
}

int gabo_nested_sphere_reconstruction(const double* x_data, const double* x_subsphere, const double* frames, const double* distances,
                                      double* cost, double* grad, int64_t P, int64_t N, int D, int levels, void* workspace,
                                      size_t workspace_bytes, gabo_stream_t stream) {
    if (D < 2 || levels < 1 || levels > D - 1) return GABO_ERR_DIM;
    const int lat = D - levels;
    const size_t lds = ((size_t)levels * lat + (size_t)levels * (levels - 1) / 2 + 2 * (size_t)D + 2 * (size_t)levels + ((size_t)gabo::chain_offset(D, levels) + 2 * (size_t)levels)) * sizeof(double);
    if (lds > 160 * 1024) return GABO_ERR_DIM;
    if (P < 0 || N < 0) return GABO_ERR_ARG;
    if (P == 0) return GABO_OK;
    if (!cost) return GABO_ERR_ARG;
    if (N == 0) {
        hipMemsetAsync(cost, 0, (size_t)P * sizeof(double), (hipStream_t)stream);
        if (grad) hipMemsetAsync(grad, 0, (size_t)P * levels * sizeof(double), (hipStream_t)stream);
        return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
    }
    if (!x_data || !x_subsphere || !frames || !distances || !workspace) return GABO_ERR_ARG;
    if (P * N > 0x7fffffffLL || workspace_bytes < gabo_nested_sphere_reconstruction_workspace_bytes(P, N, D, levels)) return GABO_ERR_ARG;
    const size_t counters = ((size_t)P * sizeof(int) + 15) / 16 * 16;
    int* cnt = static_cast<int*>(workspace);
    double* records = reinterpret_cast<double*>(static_cast<char*>(workspace) + counters);
    hipMemsetAsync(cnt, 0, counters, (hipStream_t)stream);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(gabo::nested_sphere_reconstruction_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GABO_ERR_LAUNCH;
    hipLaunchKernelGGL(gabo::nested_sphere_reconstruction_kernel, dim3((unsigned)(P * N)), dim3(64), lds, (hipStream_t)stream, x_data, x_subsphere, frames,
                       distances, cost, grad, (int)P, (int)N, D, levels, cnt, records);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

size_t gabo_nested_sphere_fit_workspace_bytes(int64_t n, int D, int levels) {
    if (n < 1 || D < 2 || levels < 1 || levels > D - 1) return 0;
    return gabo::SphereFitLayout(n, D, levels).total;
}

int gabo_nested_sphere_fit_evaluate(const double* x, const double* y, const double* axes_host, const double* distances_host, int64_t n, int D,
                                    int levels, double beta, double outputscale, double noise, double mean, int want_grad, double* out_host,
                                    void* workspace, size_t workspace_bytes, double* pinned, size_t pinned_doubles, gabo_stream_t stream) {
    if (D < 2 || D > 4096 || levels < 1 || levels > D - 1) return GABO_ERR_DIM;
    if (n < 1 || n > GABO_GP_MLL_LARGE_MAX_N) return GABO_ERR_DIM;
    const int lat = D - levels;
    if (lat > 64) return GABO_ERR_DIM;
    const size_t total = (size_t)gabo::chain_offset(D, levels);
    if (!x || !y || !axes_host || !distances_host || !out_host || !workspace || !pinned || !(beta > 0.0)) return GABO_ERR_ARG;
    const gabo::SphereFitLayout lay(n, D, levels);
    if (workspace_bytes < lay.total || pinned_doubles < 2 * total + levels + 7) return GABO_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    char* base = static_cast<char*>(workspace);
    auto at = [base](size_t off) { return reinterpret_cast<double*>(base + off); };
    double* d_axes = at(lay.axes);             // [axes (total) | distances (levels)]
    double* d_dists = d_axes + total;
    double* d_out = at(lay.out);
    int* counter = reinterpret_cast<int*>(base + lay.counters);
    std::memcpy(pinned, axes_host, sizeof(double) * total);
    std::memcpy(pinned + total, distances_host, sizeof(double) * levels);
    if (hipMemcpyAsync(d_axes, pinned, sizeof(double) * (total + levels), hipMemcpyHostToDevice, s) != hipSuccess) return GABO_ERR_LAUNCH;
    hipLaunchKernelGGL(gabo::nested_sphere_frames_kernel, dim3((unsigned)levels), dim3(64), 0, s, d_axes, at(lay.frames), D, levels);
    const size_t nfr = ((size_t)gabo::chain_offset(D, levels) + 2 * (size_t)levels);
    const size_t lds_fwd = (size_t)(D + levels) * sizeof(double), lds_bwd = ((size_t)(2 * D + levels) + nfr) * sizeof(double);
    if (!gabo::chain_lds_ok(gabo::nested_sphere_project_backward_kernel, lds_bwd)) return GABO_ERR_DIM;
    hipLaunchKernelGGL(gabo::nested_sphere_project_kernel, dim3((unsigned)n), dim3(64), lds_fwd, s, x, at(lay.frames), d_dists,
                       at(lay.z), want_grad ? at(lay.store) : nullptr, D, levels);
    if (hipGetLastError() != hipSuccess) return GABO_ERR_LAUNCH;
    int rc = gabo_sphere_pairwise(at(lay.z), at(lay.z), at(lay.kb), 1, n, n, lat, 0, 0, beta, GABO_OUT_GAUSSIAN, 0, stream);
    if (rc != GABO_OK) return rc;
    double* wm = want_grad ? at(lay.wm) : nullptr;
    if (n <= GABO_GP_MLL_MAX_N)
        rc = gabo_gp_mll_gram(at(lay.kb), y, n, outputscale, noise, mean, d_out, wm, stream);
    else
        rc = gabo_gp_mll_large(at(lay.kb), y, n, 0.0, outputscale, noise, mean, 1, d_out, wm, base + lay.mll, gabo_gp_mll_large_workspace_bytes(n), stream);
    if (rc != GABO_OK) return rc;
    size_t out_doubles = 6;
    if (want_grad) {
        if (hipMemsetAsync(counter, 0, sizeof(int), s) != hipSuccess) return GABO_ERR_LAUNCH;
        rc = gabo::fit_gram_adjoint_launch(at(lay.kb), wm, at(lay.gs), d_out, at(lay.partial), counter, n, 0.5 * outputscale, beta, lay.blocks_gram, s);
        if (rc != GABO_OK) return rc;
        hipLaunchKernelGGL(gabo::sphere_gram_adjoint_kernel, dim3((unsigned)n), dim3(64), (size_t)(65 * lat) * sizeof(double), s, at(lay.z), at(lay.gs),
                           at(lay.gz), (int)n, lat, beta);
        hipLaunchKernelGGL(gabo::nested_sphere_project_backward_kernel, dim3((unsigned)n), dim3(64), lds_bwd, s, at(lay.store),
                           at(lay.frames), d_dists, at(lay.gz), at(lay.partial), D, levels);
        hipLaunchKernelGGL(gabo::nested_sphere_axes_adjoint_kernel, dim3((unsigned)levels), dim3(64), (size_t)D * sizeof(double), s, d_axes, at(lay.frames),
                           at(lay.partial), d_out + 7, (int)n, D, levels);
        if (hipGetLastError() != hipSuccess) return GABO_ERR_LAUNCH;
        out_doubles = 7 + total;
    }
    double* h_out = pinned + total + levels;
    if (hipMemcpyAsync(h_out, d_out, sizeof(double) * out_doubles, hipMemcpyDeviceToHost, s) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return GABO_ERR_LAUNCH;
    std::memcpy(out_host, h_out, sizeof(double) * out_doubles);
    if (!want_grad) {
        out_host[6] = 0.0;
        std::memset(out_host + 7, 0, sizeof(double) * total);
    }
    return GABO_OK;
}

}  // extern "C"
