// Sphere pairwise kernel matrix on gfx950:  out_ij = f(acos(clamp(<x1_i, x2_j>, -1+1e-15, 1-1e-15)))
// replaces kernel_utils/kernels_sphere.py:71-94,118-134 + Riemannian_utils/sphere_utils_torch.py:12-55.
//
// The reference materialises two N1 x N2 x dim tensors and a degenerate bmm; here the inner products run as fp64 MFMA tiles
// (16 x 16 x 4) and each lane finishes its 16 results in registers.  HBM traffic = the output matrix (8 B per pair).  Measured at
// N = 4096, dim 10 (tools/ab_sphere.py, tools/sphere_clocks.py): 39-40 us; MFMA + stores alone 24 us, everything but the stores 38 us.
// What bound the 43-us version was not the epilogue but the memory counter (operand loads waiting behind the result stores, see the kernel).
#include "gabo_device.hpp"
#include "gabo_mirror.hpp"
#include "gabo_exp_tab256.hpp"
#include "gabo_sphere_pw_table.hpp"
#include "gabo_sphere_ktab.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {


// ---- Gaussian epilogue: K = exp(-beta acos(c)^2) without acos, in ~45 fp64 VALU instructions ------------------------------
// With z = (1 - |c|) / 2 = sin^2(phi / 2), phi = acos|c| = 2 asin(sqrt z):  (phi / 2)^2 = z P(z),  P(z) = asin(sqrt z)^2 / z analytic on
// z in [0, 1/2] (degree-17 polynomial, 4.3e-16, tools/sim/fit_sphere_poly2.py - the same accuracy as squaring a 2.2e-16 fit of
// asin(sqrt z) / sqrt z, one multiplication less).  Then
//     c >= 0:  theta^2 = phi^2 = 4 z P                        (no square root, no acos)
//     c <  0:  theta   = pi - phi,  theta^2 = phi^2 - 2 pi sqrt(phi^2) + pi^2   (the c >= 0 value + u t, u = 1, t = pi^2 - 2 pi phi)
// with u = 0 / 1 built from the sign bit of c (two 32-bit instructions; a compare + 64-bit select costs 13 cycles), and the reference's clamp of c
// to [-1+1e-15, 1-1e-15] (sphere_utils_torch.py:53) is the lower bound z >= kSphZmin.
// exp(-y), y >= 0: -y = (256 e + j) ln2/256 + r, |r| <= ln2/512: 2^e * 2^(j/256) (LDS table) * (1 + r + ... + r^4/24)  (3.8e-17).
// Against exp(-beta arccos(clip(c))^2) in 60-digit arithmetic this is as accurate as the numpy oracle itself (5e-15 vs 3.6e-15
// relative at beta = 1.3: the conditioning of exp(-beta theta^2), not the approximation; model: fit_sphere_poly2.py model 17).
constexpr int kSphWDeg = 17;
#define GABO_SPH_P_COEFFS                                                                                                              \
    0.9999999999999998, 0.3333333333336239, 0.17777777771542846, 0.11428571957076086, 0.081269605781923, 0.061574423044727364,         \
    0.048599889486560285, 0.04118585704847234, 0.02076148720348778, 0.11067290731777447, -0.3792345962104177, 1.5080499520508546,      \
    -4.072870215777421, 8.329550989544446, -12.110677571177652, 12.080701723504074, -7.375868506853218, 2.1474532497337413
__constant__ double kSphW[kSphWDeg + 1] = {GABO_SPH_P_COEFFS};
// [0..3] exp_neg_tab256: ln2/256 head and tail, 256/ln2, 1/6; [4] 1/24 (kept in a VGPR); [5] 1.5 2^52 (VGPR); [6] pi, [7] z of the clamp
__constant__ double kSphC[8] = {0.010830424695086549 / 4, 1.162596423439437e-12 / 4, 92.33248261689366 * 4, 1.0 / 6.0, 1.0 / 24.0,
                                6755399441055744.0, 3.14159265358979311600e+00, 4.996003610813204e-16};

// Host-prepared coefficients of the usual case (0 < beta < 1000, `SCALED`): P~_k = 4 beta P_k, so that z P~(z) IS beta phi^2 and the
// multiplication by 4 beta disappears from the epilogue (passed by value: kernel arguments are read through the scalar cache like the
// __constant__ table).  `a`, `b`: 2 pi sqrt(beta) and beta pi^2 of the c < 0 branch.
struct SphPoly {
    double w[kSphWDeg + 1], a, b;
};
static const double kSphWHost[kSphWDeg + 1] = {GABO_SPH_P_COEFFS};

struct SphGauss {     // wave-uniform: loaded once per kernel through the scalar cache, lives in SGPRs
    double w[kSphWDeg + 1], c[8], neg_4beta, four_pi_beta, neg_beta_pi2;
    double w_top, e_top, magic, t0;   // VGPR copies: the leading coefficient of each Horner chain, 1.5 2^52 and the addend of `t` (a VALU
                                      // instruction reads ONE scalar operand)
    template <bool SCALED>
    __device__ __forceinline__ static SphGauss load(double beta, const SphPoly& P) {
        SphGauss t;
        static_for<kSphWDeg + 1>([&](auto i) { t.w[decltype(i)::value] = SCALED ? P.w[decltype(i)::value] : kSphW[decltype(i)::value]; });
        static_for<8>([&](auto i) { t.c[decltype(i)::value] = kSphC[decltype(i)::value]; });
        if constexpr (SCALED) t.c[2] = -t.c[2];           // exp_of_minus_tab256
        t.neg_4beta = -4.0 * beta;
        t.four_pi_beta = SCALED ? -P.a : 4.0 * t.c[6] * beta;
        t.neg_beta_pi2 = SCALED ? P.b : -beta * (t.c[6] * t.c[6]);
        t.w_top = t.w[kSphWDeg];
        t.e_top = t.c[4];
        t.magic = t.c[5];
        t.t0 = t.neg_beta_pi2;
        asm volatile("" : "+v"(t.w_top), "+v"(t.e_top), "+v"(t.magic), "+v"(t.t0));
        return t;
    }
};

// 1.0 where c < 0, 0.0 elsewhere, from the sign bit
__device__ __forceinline__ double sph_neg_indicator(double ip) {
    return __hiloint2double((__double2hiint(ip) >> 31) & 0x3FF00000, 0);
}

// SCALED = false: any beta (unscaled coefficients, exp with its argument clamp)
template <bool SCALED>
__device__ __forceinline__ double sphere_gauss_finish(double ip, const SphGauss& g, const double* __restrict__ tab) {
    double z = max_raw(__builtin_fma(-0.5, __builtin_fabs(ip), 0.5), g.c[7]);
    double w = g.w_top;
    static_for<kSphWDeg>([&](auto i) { w = __builtin_fma(w, z, g.w[kSphWDeg - 1 - decltype(i)::value]); });
    double q = z * w;                                         // (phi / 2)^2, times 4 beta when SCALED
    // c < 0: t = +-(beta pi^2 - 2 pi beta phi) on top of the c >= 0 value (theta >= pi/2 there: the sum loses at most two bits)
    double t = __builtin_fma(sqrt_nz_cubic(q), g.four_pi_beta, g.t0);
    double u = sph_neg_indicator(ip);
    if constexpr (SCALED) {
        // y = beta theta^2 >= 0; y <= beta (pi^2 + eps) < 1e4: no clamp in the exp
        return exp_of_minus_tab256_magic(__builtin_fma(u, t, q), g.c, g.e_top, g.magic, tab);
    } else {
        return exp_neg_tab256<true>(__builtin_fma(u, t, q * g.neg_4beta), g.c, g.e_top, tab);
    }
}

// ---- round 3: the same value from a piecewise table, 31 + v_rsq_f64 instead of 45 + v_rsq_f64 instructions per output -----------------
// v = sqrt((1 + c) / 2) = cos(theta / 2) in [2.2e-8, 1] serves BOTH signs of c: theta^2 = 4 acos(v)^2 is analytic on all of [0, 1] (acos^2 is
// regular at v = 1, and c = -1, where theta^2 has a square-root singularity as a function of c, is the ordinary point v = 0; the only
// singularity, v = -1, is a full unit away).  So there is no c < 0 correction, no sign handling, and a degree-6 polynomial per slot of width
// 1/64 replaces the degree-17 chain.  Slot = round(64 v) by the magic-number FMA (the index appears in the low mantissa word).
// The block's copy of the table (LDS, 65 rows x 10 doubles) is pre-multiplied by -beta 256 / ln2: the Horner chain (6 FMAs, coefficients by
// three ds_read_b128 and one ds_read_b64 - the LDS pipe, not the vector pipe) delivers Y = -beta theta^2 in units of ln2/256, the exp argument
// reduction is the exact subtraction Y - rint(Y) (no head/tail product), and exp(r ln2/256) - 1 is a degree-4 polynomial with the scale inside its
// coefficients.  Instructions per output: 3 (clamped q) + v_rsq_f64 + 5 + 3 (slot, t) + 1 (address) + 6 + 12 (exp) = 30 + v_rsq_f64 (13.7 cycles).
// Accuracy: tools/sim/fit_sphere_piecewise.py model 6 - 2.9e-15 against 60-digit arithmetic at beta = 1.29 (numpy's own 3.1e-15).
// A NaN inner product does not propagate through this chain (v_max_f64 / v_min_f64 return the bound), nor through the clamp of the other
// epilogues; the reference's clamp -> acos -> exp returns NaN (sphere_utils_torch.py:53-55).  The pair kernel therefore repairs such
// entries behind the epilogue (`nan_fixup` below), at the price of one compare per operand and chunk instead of two instructions per output.
struct SphPwRegs {
    double qmin, qmax, c1, c2, c3, magic;    // SGPRs
    double c4, scale;                        // VGPRs (a VALU instruction reads ONE scalar operand: these share an FMA with c3 resp. the magic number;
                                             // as the scalar one of its FMA the magic number is an addend the compiler cannot turn into v_mov + v_fmac)
    __device__ __forceinline__ static SphPwRegs load();
};
// from constant memory (scalar loads -> SGPRs; as literals the compiler keeps them in VGPRs and copies one per v_fmac):
// [0], [1] the reference's clamp of c to [-1 + 1e-15, 1 - 1e-15] (sphere_utils_torch.py:53) stated for q = (1 + c) / 2; [2] slots per unit
// of v; [3..6] l^k / k!, l = ln2 / 256; [7] 1.5 2^52
#define GABO_SPH_L 0.0027076061740622863
__constant__ double kSphPwC[8] = {4.996003610813204e-16, 1.0 - 4.996003610813204e-16, kSphPwScale, GABO_SPH_L, GABO_SPH_L* GABO_SPH_L / 2.0,
                                  GABO_SPH_L* GABO_SPH_L* GABO_SPH_L / 6.0, GABO_SPH_L* GABO_SPH_L* GABO_SPH_L* GABO_SPH_L / 24.0,
                                  6755399441055744.0};
__device__ __forceinline__ SphPwRegs SphPwRegs::load() {
    SphPwRegs t;
    t.qmin = kSphPwC[0];
    t.qmax = kSphPwC[1];
    t.scale = kSphPwC[2];
    t.c1 = kSphPwC[3];
    t.c2 = kSphPwC[4];
    t.c3 = kSphPwC[5];
    t.c4 = kSphPwC[6];
    t.magic = kSphPwC[7];
    asm volatile("" : "+s"(t.qmin), "+s"(t.qmax), "+s"(t.c1), "+s"(t.c2), "+s"(t.c3), "+s"(t.magic));
    asm volatile("" : "+v"(t.c4), "+v"(t.scale));
    return t;
}

__device__ __forceinline__ double min_raw(double x, double bound_uniform) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(bound_uniform));
    return r;
}

__device__ __forceinline__ double sphere_gauss_finish_pw(double ip, const SphPwRegs& g, const double* __restrict__ pw,
                                                         const double* __restrict__ tab) {
    const double q = min_raw(max_raw(__builtin_fma(0.5, ip, 0.5), g.qmin), g.qmax);
    const double v = sqrt_nz_cubic(q);
    const double kf = __builtin_fma(v, g.scale, g.magic);
    const double kd = kf - g.magic;
    const double t = __builtin_fma(v, g.scale, -kd);
    // the slot's seven coefficients: three ds_read_b128 + one ds_read_b64 (conflict-free 4 / 2 LDS cycles each; the two-address
    // ds_read2st64_b64 the compiler merges a [coefficient][slot] layout into runs at half that rate and made the LDS pipe the bottleneck:
    // 49 instead of 41 us).  Byte offset by the 24-bit multiply: v_mul_lo_u32, which a plain `slot * 10` compiles to, is quarter rate.
    typedef double pw_v2d __attribute__((ext_vector_type(2)));
    const char* rowb = reinterpret_cast<const char*>(pw) + __umul24((unsigned)__double2loint(kf), (unsigned)(kSphPwStride * sizeof(double)));
    const pw_v2d* row = reinterpret_cast<const pw_v2d*>(rowb);
    static_assert(kSphPwDeg == 6, "three coefficient pairs and one single");
    const double c6 = reinterpret_cast<const double*>(rowb)[6];
    const pw_v2d c45 = row[2], c23 = row[1], c01 = row[0];
    double w = __builtin_fma(c6, t, c45[1]);
    w = __builtin_fma(w, t, c45[0]);
    w = __builtin_fma(w, t, c23[1]);
    w = __builtin_fma(w, t, c23[0]);
    w = __builtin_fma(w, t, c01[1]);
    w = __builtin_fma(w, t, c01[0]);
    const double km = w + g.magic;
    const double k = km - g.magic;
    const double r = w - k;
    double p = __builtin_fma(r, g.c4, g.c3);
    p = __builtin_fma(p, r, g.c2);
    p = __builtin_fma(p, r, g.c1);
    p = p * r;                                                 // exp(r ln2/256) - 1
    const int ki = __double2loint(km);
    const double e = tab[ki & 255];
    return __builtin_ldexp(__builtin_fma(e, p, e), ki >> 8);
}

// ---- round 4: the KERNEL VALUE itself from a piecewise table: no exp per output, 17 + v_rsq_f64 instructions instead of 31 + v_rsq_f64 -------
// K(v) = exp(-beta Theta(v)), Theta = 4 acos^2, is as smooth in v = cos(theta / 2) as Theta (the only singularity, v = -1, is a unit away from
// [0, 1]); with 1024 slots of width 1/1024 a degree-5 polynomial per slot reproduces it to the accuracy of the inputs for beta <= 4
// (tools/sim/gen_sphere_ktab.py: 2.0e-15 against 60-digit arithmetic at beta = 1.29, numpy's own acos^2 / exp chain 3.4e-15).  The table depends on
// beta, so every block builds its copy at launch (49 KB of LDS, one slot per thread of a 1024-thread block): from the beta-independent Taylor
// coefficients theta_k(s) of Theta at the slot centres (csrc/gabo_sphere_ktab.hpp, 60-digit generation) by the power-series recurrence of
// exp(polynomial),  e_0 = exp(p_0),  e_m = (1/m) sum_k k p_k e_(m-k),  p_k = -beta theta_k,  to degree 6, the t^6 term folded into the lower ones
// by Chebyshev economisation on [-1/2, 1/2] (t^6 ~ (768 t^4 - 72 t^2 + 1) / 2048).  ~75 instructions per thread once, against 14 saved on each of
// its 64 outputs.  The clamp, the square root and the magic-number slot selection are those of the round-3 epilogue.
constexpr int kSphKtStride = 6;      // doubles per LDS row: three ds_read_b128 (48 bytes; rows s and s + 16 k share LDS banks)
constexpr double kSphKtMaxBeta = 4.0;
struct SphKtRegs {
    double qmin, qmax, magic;     // SGPRs
    double scale;                 // VGPR (shares an FMA with the magic number)
    __device__ __forceinline__ static SphKtRegs load() {
        SphKtRegs t;
        t.qmin = kSphPwC[0];
        t.qmax = kSphPwC[1];
        t.magic = kSphPwC[7];
        t.scale = (double)kSphKtScale;
        asm volatile("" : "+s"(t.qmin), "+s"(t.qmax), "+s"(t.magic));
        asm volatile("" : "+v"(t.scale));
        return t;
    }
};

// the six coefficients of slot s (one thread): see above.  The base row is loaded separately (the first one at the very top of the kernel,
// together with the operands: a second dependent round trip to L2 / HBM in front of the block's barrier was a third of the prologue)
typedef double kt_v2d __attribute__((ext_vector_type(2)));
struct SphKtBaseRow {
    kt_v2d b01, b23, b45, b67;
    __device__ __forceinline__ static SphKtBaseRow load(int s) {
        const kt_v2d* b = reinterpret_cast<const kt_v2d*>(kSphKtBase + s * kSphKtBaseStride);
        return SphKtBaseRow{b[0], b[1], b[2], b[3]};
    }
};
__device__ __forceinline__ void sphere_kt_build_row(int s, const SphKtBaseRow& br, double beta, double* __restrict__ kt) {
    const kt_v2d b01 = br.b01, b23 = br.b23, b45 = br.b45, b67 = br.b67;
    const double nb = -beta;
    const double p0 = nb * b01[0];
    const double q1 = nb * b01[1], q2 = (2.0 * nb) * b23[0], q3 = (3.0 * nb) * b23[1], q4 = (4.0 * nb) * b45[0], q5 = (5.0 * nb) * b45[1],
                 q6 = (6.0 * nb) * b67[0];
    const double e0 = exp(p0);
    const double e1 = q1 * e0;
    const double e2 = 0.5 * __builtin_fma(q1, e1, q2 * e0);
    const double e3 = (1.0 / 3.0) * __builtin_fma(q1, e2, __builtin_fma(q2, e1, q3 * e0));
    const double e4 = 0.25 * __builtin_fma(q1, e3, __builtin_fma(q2, e2, __builtin_fma(q3, e1, q4 * e0)));
    const double e5 = 0.2 * __builtin_fma(q1, e4, __builtin_fma(q2, e3, __builtin_fma(q3, e2, __builtin_fma(q4, e1, q5 * e0))));
    const double e6 = (1.0 / 6.0) * __builtin_fma(q1, e5, __builtin_fma(q2, e4, __builtin_fma(q3, e3, __builtin_fma(q4, e2, __builtin_fma(q5, e1, q6 * e0)))));
    kt_v2d* row = reinterpret_cast<kt_v2d*>(kt + s * kSphKtStride);
    row[0] = kt_v2d{__builtin_fma(e6, 1.0 / 2048.0, e0), e1};
    row[1] = kt_v2d{__builtin_fma(e6, -72.0 / 2048.0, e2), e3};
    row[2] = kt_v2d{__builtin_fma(e6, 768.0 / 2048.0, e4), e5};
}

__device__ __forceinline__ double sphere_gauss_finish_kt(double ip, const SphKtRegs& g, const double* __restrict__ kt) {
    const double q = min_raw(max_raw(__builtin_fma(0.5, ip, 0.5), g.qmin), g.qmax);
    const double v = sqrt_nz_cubic(q);
    const double kf = __builtin_fma(v, g.scale, g.magic);
    const double kd = kf - g.magic;
    const double t = __builtin_fma(v, g.scale, -kd);
    const char* rowb = reinterpret_cast<const char*>(kt) + __umul24((unsigned)__double2loint(kf), (unsigned)(kSphKtStride * sizeof(double)));
    const kt_v2d* row = reinterpret_cast<const kt_v2d*>(rowb);
    const kt_v2d c45 = row[2], c23 = row[1], c01 = row[0];
    double w = __builtin_fma(c45[1], t, c45[0]);
    w = __builtin_fma(w, t, c23[1]);
    w = __builtin_fma(w, t, c23[0]);
    w = __builtin_fma(w, t, c01[1]);
    return __builtin_fma(w, t, c01[0]);
}

// true for NaN, +-inf and magnitudes whose products could overflow: operands of an inner product that may come out NaN
__device__ __forceinline__ bool sph_suspect(double v) { return !(__builtin_fabs(v) < 1e150); }

template <int MODE>
__device__ __forceinline__ double sphere_finish(double ip, double beta, const MathRegs& mt) {
    const double lo = -1.0 + 1e-15, hi = 1.0 - 1e-15;  // sphere_utils_torch.py:53
    double c = __builtin_fmin(__builtin_fmax(ip, lo), hi);
    double dist = acos_fast(c, mt);                       // :55
    if constexpr (MODE == GABO_OUT_DISTANCE) return dist;
    if constexpr (MODE == GABO_OUT_LAPLACE) return exp_neg(-(dist * beta), mt);
    return exp_neg(-((dist * dist) * beta), mt);          // kernels_sphere.py:91-93
}

typedef double sph_v4d __attribute__((ext_vector_type(4)));

// grid.x: (row chunk, column group), column group fastest; grid.y: batch.  A block owns `rows` = 16 * chunks rows x blockDim.x columns.
// The inner products <x1_i, x2_j> are a GEMM with K = dim: v_mfma_f64_16x16x4_f64.  A wave owns 64 columns = four 16 x 16 MFMA tiles
// per 16-row chunk and walks the block's row chunks; per K step of 4 a lane holds ONE double of x1 (A[i = lane & 15][k = lane >> 4])
// and one of each x2 tile (B[k = lane >> 4][j = lane & 15]) - operands are read in exactly the fragment layout, no transposes (where they
// come from: KS below).  Result layout of the f64 MFMA: register r of lane l is row (l >> 4) + 4 r, column l & 15 - a wave store writes
// four full 128-byte lines.
// Issue budget: tools/ubench_mfma_f64.hip shows the f64 matrix pipe and the f64 vector pipe do NOT overlap on gfx950 (MFMA alone 77.8,
// v_fma_f64 alone 60.6, both together 64.9 TFLOP/s), so per output the MFMA costs 12 issue slots beside the ~48 instructions of the
// epilogue (27.5 us of the 38-us kernel span; the rest is prologue, the lone-wave tail and store interference: tools/sphere_clocks.py).
// KS > 0: dim <= 4 KS <= 16 - the wave's x2 fragments stay in registers for all of its row chunks and the block's x1 rows are copied to
// LDS once, so that the chunk loop contains NO global load.  This is about the memory counter, not about load latency: vmcnt retires in
// issue order, loads and stores alike, so operand loads issued after a chunk's 16 stores cannot be waited for without also waiting for
// those stores to be acknowledged by L2 - under a 3-5 TB/s write stream that stalled every wave once per chunk (measured at N = 16384:
// 633 us with the loads behind the stores, 478 us with the stores removed, 369 us for the stores alone).  LDS reads count in lgkmcnt.
// KS = 0: any dim, operands loaded per chunk.
constexpr int kSphMaxChunks = 8;
// four waves per SIMD (128 VGPRs): neither the timestamps of the development build nor the 16 independent table-driven epilogues of a
// chunk (which the scheduler would otherwise interleave into 160 registers) may cost the fourth wave
#ifndef GABO_SPH_PW_WAVES
#define GABO_SPH_PW_WAVES 4
#endif
#ifndef GABO_SPH_PW_GROUP
#define GABO_SPH_PW_GROUP 1
#endif
#ifndef GABO_SPH_PW_BARRIER
#define GABO_SPH_PW_BARRIER 1
#endif
#ifndef GABO_SPH_PW_EVERY      /* epilogues the scheduler may interleave between two scheduling barriers */
#define GABO_SPH_PW_EVERY 1
#endif
#ifdef GABO_SPH_CLOCKS
#define GABO_SPH_BOUNDS __launch_bounds__(KT ? 1024 : 256) __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define GABO_SPH_BOUNDS __launch_bounds__(KT ? 1024 : 256) __attribute__((amdgpu_waves_per_eu((PW || KT) ? GABO_SPH_PW_WAVES : 1, 4)))
#endif
#ifndef GABO_SPH_KT_EVERY      /* epilogues of the kernel-value table the scheduler may interleave between two scheduling barriers */
#define GABO_SPH_KT_EVERY 2
#endif
constexpr int kSphKtChunks = 4;      // 16-row chunks per block of the KT variant (its LDS copy of the x1 rows)
template <int MODE, bool SCALED = false, int KS = 0, bool NT = false, bool PW = false, bool KT = false>
__global__ GABO_SPH_BOUNDS void sphere_pairwise_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                              double* __restrict__ out, int64_t n1, int64_t n2, int dim,
                                                              int64_t s1, int64_t s2, int col_blocks, int row_chunks, int chunks,
                                                              double beta, int flags, SphPoly poly, const double* __restrict__ ktab_g) {
    static_assert(!KT || (KS > 0 && !PW && MODE == GABO_OUT_GAUSSIAN), "the kernel-value table serves the Gaussian mode with the operands in registers");
    __shared__ double tab[KT ? 1 : 256];
    __shared__ double xa[KS > 0 ? 16 * (KT ? kSphKtChunks : kSphMaxChunks) * 4 * KS : 1];
    __shared__ __attribute__((aligned(16))) double pw[PW ? kSphPwSlots * kSphPwStride : 2];
    __shared__ __attribute__((aligned(16))) double kt[KT ? kSphKtSlots * kSphKtStride : 2];
    const int tid = threadIdx.x;
    SphGauss g;       // requested first: the scalar loads of the coefficients travel together with the kernel arguments
    SphPwRegs gp;
    SphKtRegs gk;
    MathRegs mt;
    if constexpr (KT) gk = SphKtRegs::load();
    else if constexpr (PW) gp = SphPwRegs::load();
    else if constexpr (MODE == GABO_OUT_GAUSSIAN) g = SphGauss::load<SCALED>(beta, poly);
    else mt = MathRegs::load();
    SphKtBaseRow kbase;
    kt_v2d krow[3];
    if constexpr (KT) {
        // the caller's table for this beta (gabo_sphere_ktable_build), if any: the thread's row travels with the operands and goes to LDS as it
        // is - no exp, no recurrence in the prologue; otherwise the beta-independent base row the table is built from
        const int srow = tid < kSphKtSlots ? tid : kSphKtSlots - 1;       // (blocks are at most kSphKtSlots threads)
        if (ktab_g) {
            const kt_v2d* src = reinterpret_cast<const kt_v2d*>(ktab_g + srow * kSphKtStride);
            krow[0] = src[0], krow[1] = src[1], krow[2] = src[2];
        } else {
            kbase = SphKtBaseRow::load(srow);
        }
    }
    constexpr int kPwN = kSphPwSlots * kSphPwStride, kPwFirst = (kPwN + 255) / 256;       // 650 entries: three per thread of a 256-thread block
    double pwv[kPwFirst];
    if constexpr (PW) {
        // the piecewise table: its first round of loads (all of it for a 256-thread block) is requested with everything else
        static_for<kPwFirst>([&](auto ee) {
            const int k = tid + decltype(ee)::value * (int)blockDim.x;
            pwv[decltype(ee)::value] = kSphPwTab[k < kPwN ? k : kPwN - 1];
        });
    }
    auto pw_store = [&]() {      // ... and scaled by -beta 256 / ln2 on its way to LDS
        if constexpr (PW) {
            const double sc = beta * -369.3299304675746;
            static_for<kPwFirst>([&](auto ee) {
                const int k = tid + decltype(ee)::value * (int)blockDim.x;
                if (k < kPwN) pw[k] = pwv[decltype(ee)::value] * sc;
            });
            for (int k = kPwFirst * (int)blockDim.x + tid; k < kPwN; k += (int)blockDim.x) pw[k] = kSphPwTab[k] * sc;      // blocks of < 256 threads
        }
    };
#ifdef GABO_SPH_CLOCKS    /* development: per-wave timestamps (100 MHz) into the output buffer; build with GABO_SPH_PROBE=2 (no result stores) */
    const uint64_t clk_start = __builtin_amdgcn_s_memrealtime();
    const uint64_t cyc_start = __builtin_amdgcn_s_memtime();
#endif
    const int rows = 16 * chunks;
    uint32_t cg, rc;
    if (flags & GABO_SYMMETRIC) {       // x1 is x2: only tiles touching the upper triangle exist (see spd_pairwise.hip)
        uint32_t t = blockIdx.x;
        cg = 0;
        for (;;) {
            uint32_t cnt = (uint32_t)sym_chunks_of(cg, blockDim.x, rows, row_chunks);
            if (t < cnt) break;
            t -= cnt;
            ++cg;
        }
        rc = t;
    } else {
        rc = blockIdx.x / (uint32_t)col_blocks;
        cg = blockIdx.x - rc * (uint32_t)col_blocks;
    }
    const int64_t b = blockIdx.y;
    const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
    // first column of this wave (>= n2: the wave only helps with the copies).  The wave index goes through v_readfirstlane so that the
    // compiler KNOWS j0 is wave-uniform: the result stores then take their row base from SGPRs (`global_store ... v_off, v_data, s[base]`)
    // instead of one 64-bit vector addition per store
    const int64_t j0 = (int64_t)cg * blockDim.x + 64 * __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* pb[4];
    static_for<4>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        int64_t jb = j0 + 16 * t + li;
        jb = jb < n2 ? jb : n2 - 1;                   // out-of-range rows / columns recompute the last one and are not stored
        pb[t] = x2 + b * s2 + jb * dim;
    });
    constexpr int KSR = KS > 0 ? KS : 1;
    double bfrag[KSR][4];
    int kcs[KSR];
    if constexpr (KS > 0) {
        // every global load of the kernel is requested here, before anything waits: the wave's x2 fragments, then the block's rows of x1
        // (rows * dim consecutive doubles; past the end of the set: its last entry, never stored) and the exp table on their way to LDS
        static_for<KS>([&](auto ss) {
            constexpr int sidx = decltype(ss)::value;
            const int kk = 4 * sidx + lk;
            kcs[sidx] = kk < dim ? kk : dim - 1;
            static_for<4>([&](auto tt) { bfrag[sidx][decltype(tt)::value] = pb[decltype(tt)::value][kcs[sidx]]; });
        });
        // K is padded to 4 KS with zeros on BOTH sides (0 * 0: an infinite entry of the last valid column must not meet the zero of the other
        // operand); once per wave
        static_for<4>([&](auto tt) {
            bfrag[KS - 1][decltype(tt)::value] = 4 * (KS - 1) + lk >= dim ? 0.0 : bfrag[KS - 1][decltype(tt)::value];
        });
        const int64_t first = (int64_t)rc * rows * dim, last = n1 * dim - 1;
        const double* src = x1 + b * s1;
        const int total = rows * dim, step = (int)blockDim.x;
        double tv = 0.0;
        if constexpr (MODE == GABO_OUT_GAUSSIAN && !KT) tv = kExp2Tab256[tid];
        for (int base = tid; base < total; base += 4 * step) {       // four loads in flight per thread (one round for 64 rows x dim <= 16)
            double stage[4];
            static_for<4>([&](auto ee) {
                const int64_t k = first + base + decltype(ee)::value * step;
                stage[decltype(ee)::value] = src[k < last ? k : last];
            });
            static_for<4>([&](auto ee) {
                const int k = base + decltype(ee)::value * step;
                if (k < total) xa[k] = stage[decltype(ee)::value];
            });
        }
        if constexpr (KT) {
            // the block's table of kernel values for this beta: one slot per thread (see sphere_kt_build_row)
            if (ktab_g) {
                kt_v2d* dst = reinterpret_cast<kt_v2d*>(kt + tid * kSphKtStride);
                dst[0] = krow[0], dst[1] = krow[1], dst[2] = krow[2];
                for (int sl = tid + step; sl < kSphKtSlots; sl += step) {
                    const kt_v2d* src = reinterpret_cast<const kt_v2d*>(ktab_g + sl * kSphKtStride);
                    kt_v2d* d2 = reinterpret_cast<kt_v2d*>(kt + sl * kSphKtStride);
                    d2[0] = src[0], d2[1] = src[1], d2[2] = src[2];
                }
            } else {
                sphere_kt_build_row(tid, kbase, beta, kt);
                for (int sl = tid + step; sl < kSphKtSlots; sl += step) sphere_kt_build_row(sl, SphKtBaseRow::load(sl), beta, kt);
            }
        } else if constexpr (MODE == GABO_OUT_GAUSSIAN) {
            tab[tid] = tv;
            for (int k = tid + step; k < 256; k += step) tab[k] = kExp2Tab256[k];
        }
        pw_store();
        __syncthreads();
    } else if constexpr (MODE == GABO_OUT_GAUSSIAN) {
        for (int k = tid; k < 256; k += blockDim.x) tab[k] = kExp2Tab256[k];
        pw_store();
        __syncthreads();
    }
    if (j0 >= n2) return;
    auto finish_raw = [&](double ip) {
#if defined(GABO_SPH_PROBE) && GABO_SPH_PROBE == 1     /* development probe: MFMA + stores only */
        return ip;
#endif
        if constexpr (KT) return sphere_gauss_finish_kt(ip, gk, kt);
        else if constexpr (PW) return sphere_gauss_finish_pw(ip, gp, pw, tab);
        else if constexpr (MODE == GABO_OUT_GAUSSIAN) return sphere_gauss_finish<SCALED>(ip, g, tab);
        else return sphere_finish<MODE>(ip, beta, mt);
    };
    // NaN inner products come out NaN, as the reference's clamp -> acos -> exp does (sphere_utils_torch.py:53-55), although the clamps of the
    // epilogues above return their bound for a NaN.  KS > 0: the operands are in registers - one compare each per chunk decides, wave-uniformly,
    // whether any inner product of the chunk CAN be NaN (NaN / inf operand, or magnitudes whose products overflow); only then the chunk's
    // tiles are formed once more and their NaN entries stored over the epilogue's values (`nan_fixup`).  KS = 0: a select per output.
    auto finish = [&](double ip) {
        const double v = finish_raw(ip);
        if constexpr (KS > 0) return v;
        else return ip != ip ? ip : v;
    };
    bool suspect_b = false;
#ifndef GABO_SPH_NO_NAN_FIXUP
    if constexpr (KS > 0) {
        static_for<KS>([&](auto ss) {
            static_for<4>([&](auto tt) { suspect_b |= sph_suspect(bfrag[decltype(ss)::value][decltype(tt)::value]); });
        });
        suspect_b = __builtin_amdgcn_ballot_w64(suspect_b) != 0;
    }
#endif
    const uint32_t loff = ((uint32_t)lk * (uint32_t)n2 + (uint32_t)li) * 8u;       // byte offset of the lane inside a 4-row group (n2 < 2^27)
#ifdef GABO_SPH_CLOCKS
    const uint64_t clk_loop = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ch = 0; ch < chunks; ++ch) {
        const int64_t i0 = (int64_t)rc * rows + 16 * ch;
        if (i0 >= n1) break;
        // x1 is x2, wave entirely left of the chunk's first row: i > j for every pair it would evaluate (and for every later chunk)
        if ((flags & GABO_SYMMETRIC) && j0 + 63 < i0) break;
        sph_v4d acc[4];
        // Stores: wave-uniform base of the chunk + a 32-bit lane offset.  A chunk that lies fully inside the matrix (and, with
        // x1 is x2, fully above the diagonal) stores without predicates.
        double* ob = out + b * n1 * n2 + i0 * n2 + j0;
        const bool inside = i0 + 16 <= n1 && j0 + 64 <= n2 && (uint64_t)n2 < (1ull << 27) &&
                            (!(flags & GABO_SYMMETRIC) || i0 + 15 <= j0);
        // finish and store tiles T0 ... T0 + NTILE - 1 of the chunk
        auto store_tiles = [&](auto t0_, auto ntile_) {
            constexpr int T0 = decltype(t0_)::value, NTILE = decltype(ntile_)::value;
            if (inside) {
                static_for<4>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    double* orow = ob + (int64_t)(4 * r) * n2;       // wave-uniform
                    static_for<NTILE>([&](auto tt) {
                        constexpr int t = T0 + decltype(tt)::value;
#if defined(GABO_SPH_PROBE) && GABO_SPH_PROBE == 2     /* development probe: everything but the stores */
                        double v_ = finish(acc[t][r]);
                        if (v_ == 12345.678) *reinterpret_cast<double*>(reinterpret_cast<char*>(orow) + loff + 128u * t) = v_;
#else
                        double* dst = reinterpret_cast<double*>(reinterpret_cast<char*>(orow) + loff + 128u * t);
                        if constexpr (NT) __builtin_nontemporal_store(finish(acc[t][r]), dst);
                        else *dst = finish(acc[t][r]);
#endif
                        // one table-driven epilogue at a time: left alone the scheduler interleaves the whole chunk (8 table reads and ~35
                        // registers per output) and the kernel no longer fits the 128 registers of four waves per SIMD
                        if constexpr (PW && GABO_SPH_PW_BARRIER && ((r * NTILE + decltype(tt)::value + 1) % GABO_SPH_PW_EVERY == 0)) __builtin_amdgcn_sched_barrier(0);
                        if constexpr (KT && ((r * NTILE + decltype(tt)::value + 1) % GABO_SPH_KT_EVERY == 0)) __builtin_amdgcn_sched_barrier(0);
                    });
                });
            } else {
                static_for<NTILE>([&](auto tt) {
                    constexpr int t = T0 + decltype(tt)::value;
                    const int64_t j = j0 + 16 * t + li;
                    static_for<4>([&](auto rr) {
                        constexpr int r = decltype(rr)::value;
                        const int64_t i = i0 + lk + 4 * r;
                        double val = finish(acc[t][r]);
                        if (i < n1 && j < n2 && (!(flags & GABO_SYMMETRIC) || i <= j)) ob[(int64_t)(lk + 4 * r) * n2 + 16 * t + li] = val;
                    });
                });
            }
        };
        if constexpr (KS > 0) {
            // K is padded to 4 KS with zeros: the padded lanes hold the last valid entry and zero it (the x2 operand was zeroed once per wave)
            double a_cur[KS];
            const double* xrow = xa + (16 * ch + li) * dim;
            static_for<KS>([&](auto ss) {
                constexpr int sidx = decltype(ss)::value;
                a_cur[sidx] = xrow[kcs[sidx]];
                if constexpr (sidx == KS - 1) a_cur[sidx] = 4 * sidx + lk >= dim ? 0.0 : a_cur[sidx];
            });
            // PW: two tiles at a time (their MFMAs, then their eight epilogues) - 16 accumulator registers live instead of 32
            constexpr int GROUP = ((PW && GABO_SPH_PW_BARRIER) || KT) ? GABO_SPH_PW_GROUP : 4;
            auto form_tiles = [&](auto t0_, auto ntile_) {
                constexpr int T0 = decltype(t0_)::value, NTILE = decltype(ntile_)::value;
                static_for<KS>([&](auto ss) {
                    constexpr int sidx = decltype(ss)::value;
                    static_for<NTILE>([&](auto tt) {
                        constexpr int t = T0 + decltype(tt)::value;
                        if constexpr (sidx == 0) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[0], bfrag[0][t], sph_v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
                        else acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[sidx], bfrag[sidx][t], acc[t], 0, 0, 0);
                    });
                });
            };
            static_for<4 / GROUP>([&](auto gg) {
                constexpr int T0 = decltype(gg)::value * GROUP;
                form_tiles(std::integral_constant<int, T0>{}, std::integral_constant<int, GROUP>{});
                store_tiles(std::integral_constant<int, T0>{}, std::integral_constant<int, GROUP>{});
                if constexpr ((PW && GABO_SPH_PW_BARRIER) || KT) __builtin_amdgcn_sched_barrier(0);
            });
            bool suspect = false;
#ifndef GABO_SPH_NO_NAN_FIXUP     /* A/B: the round-3 kernel (NaN inner products not repaired) */
            static_for<KS>([&](auto ss) { suspect |= sph_suspect(a_cur[decltype(ss)::value]); });
#endif
            if (__builtin_expect(suspect_b || __builtin_amdgcn_ballot_w64(suspect) != 0, 0)) {
                // nan_fixup (rare): the chunk's inner products once more, tile by tile; their NaN entries replace what the epilogue stored
                // (same lane, same address, program order)
                static_for<4>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    form_tiles(std::integral_constant<int, t>{}, std::integral_constant<int, 1>{});
                    const int64_t j = j0 + 16 * t + li;
                    static_for<4>([&](auto rr) {
                        constexpr int r = decltype(rr)::value;
                        const int64_t i = i0 + lk + 4 * r;
                        const double ipn = acc[t][r];
                        if (ipn != ipn && i < n1 && j < n2 && (!(flags & GABO_SYMMETRIC) || i <= j)) {
                            double* dst = ob + (int64_t)(lk + 4 * r) * n2 + 16 * t + li;
                            if constexpr (NT) __builtin_nontemporal_store(ipn, dst);
                            else *dst = ipn;
                        }
                    });
                });
            }
        } else {
            const int64_t ia = (i0 + li < n1) ? i0 + li : n1 - 1;
            const double* pa = x1 + b * s1 + ia * dim;
            // K is padded to a multiple of 4 with zeros: the padded lanes load the last valid entry (no divergent load) and zero both
            // operands (0 * 0: an infinite entry must not meet the zero of the other side).  The first step starts from a literal zero
            // accumulator (no per-chunk clearing moves).
            auto kstep = [&](int k0, auto first) {
                const int kk = k0 + lk;
                const bool ok = kk < dim;
                const int kc = ok ? kk : dim - 1;
                double a = pa[kc];
                double bv[4];
                static_for<4>([&](auto tt) { bv[decltype(tt)::value] = pb[decltype(tt)::value][kc]; });
                a = ok ? a : 0.0;
                static_for<4>([&](auto tt) { bv[decltype(tt)::value] = ok ? bv[decltype(tt)::value] : 0.0; });
                static_for<4>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    if constexpr (decltype(first)::value) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[t], sph_v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[t], acc[t], 0, 0, 0);
                });
            };
            kstep(0, std::true_type{});
            for (int k0 = 4; k0 < dim; k0 += 4) kstep(k0, std::false_type{});
            store_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        }
    }
#ifdef GABO_SPH_CLOCKS
    if ((tid & 63) == 0) {     // records live BEHIND the result matrix: the caller (tools/sphere_clocks.py) allocates 4 doubles per wave more
        const uint64_t clk_end = __builtin_amdgcn_s_memrealtime();
        const uint64_t cyc_end = __builtin_amdgcn_s_memtime();
        double* rec = out + (int64_t)gridDim.y * n1 * n2 + ((int64_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)) * 4;
        rec[0] = (double)clk_start;
        rec[1] = (double)clk_loop;
        rec[2] = (double)clk_end;
        rec[3] = (double)(cyc_end - cyc_start);       // shader cycles entry -> exit
    }
#endif
}

// diag branch: row k of x1 with row k of x2 (sphere_utils_torch.py:45-49)
__global__ __launch_bounds__(256) void sphere_diag_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                          double* __restrict__ out, int64_t batch, int64_t n, int dim,
                                                          int64_t s1, int64_t s2, double beta, int flags) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= batch * n) return;
    int64_t b = g / n, i = g - b * n;
    const double* p = x1 + b * s1 + i * dim;
    const double* q = x2 + b * s2 + i * dim;
    double acc = 0.0;
    for (int k = 0; k < dim; ++k) acc = __builtin_fma(p[k], q[k], acc);
    const MathRegs mt = MathRegs::load();
    const int mode = flags & GABO_OUT_MASK;
    const double v = mode == GABO_OUT_DISTANCE ? sphere_finish<GABO_OUT_DISTANCE>(acc, beta, mt)
                                               : (mode == GABO_OUT_LAPLACE ? sphere_finish<GABO_OUT_LAPLACE>(acc, beta, mt)
                                                                           : sphere_finish<GABO_OUT_GAUSSIAN>(acc, beta, mt));
    out[g] = acc != acc ? acc : v;      // the clamp inside sphere_finish returns its bound for a NaN; the reference's returns NaN
}

// Element-wise f^(order)(c) on a precomputed inner-product matrix c = <x1_i, x2_j>, f(c) = g(clamp(c)):
//   Gaussian g = exp(-beta acos(c)^2), Laplace g = exp(-beta acos(c)), distance g = acos(c).
// order 0/1/2 = value / first / second derivative with respect to c (zero where the clamp is active, autograd `clamp`
// semantics).  This is the differentiable path: the reference differentiates through clamp/acos/exp by autograd, twice for
// the exact Hessian-vector products of the sphere trust region (pymanopt_addons/tools/autodiff/_pytorch.py:103-116).
__global__ __launch_bounds__(256) void sphere_from_inner_kernel(const double* __restrict__ cin, double* __restrict__ out, int64_t n,
                                                                double beta, int mode, int order) {
    const double lo = -1.0 + 1e-15, hi = 1.0 - 1e-15;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
        double ip = cin[g];
        bool inside = ip >= lo && ip <= hi;
        double c = ip < lo ? lo : (ip > hi ? hi : ip);
        double th = acos(c);
        double res;
        if (order == 0) {
            res = (mode == GABO_OUT_DISTANCE) ? th : (mode == GABO_OUT_LAPLACE ? exp(-(th * beta)) : exp(-((th * th) * beta)));
        } else {
            double om = (1.0 - c) * (1.0 + c);
            double t1 = -1.0 / __builtin_sqrt(om);     // theta'
            double t2 = c * t1 / om;                   // theta'' = -c (1-c^2)^-3/2
            if (mode == GABO_OUT_DISTANCE) {
                res = order == 1 ? t1 : t2;
            } else if (mode == GABO_OUT_LAPLACE) {
                double gv = exp(-(th * beta));
                double a = -beta * t1;
                res = order == 1 ? gv * a : gv * (a * a - beta * t2);
            } else {
                double gv = exp(-((th * th) * beta));
                double a = -2.0 * beta * th * t1;
                res = order == 1 ? gv * a : gv * (a * a - 2.0 * beta * (t1 * t1 + th * t2));
            }
            if (!inside) res = 0.0;
        }
        out[g] = res;
    }
}

}  // namespace gabo

extern "C" int gabo_sphere_from_inner(const double* inner, double* out, int64_t n, double beta, int flags, int order,
                                      gabo_stream_t stream) {
    if (n < 0 || order < 0 || order > 2) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    if (!inner || !out) return GABO_ERR_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gabo::sphere_from_inner_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, inner, out, n, beta,
                       flags & GABO_OUT_MASK, order);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

namespace gabo {
// the launch condition of the kernel-value table path (sphere_gauss_finish_kt), in one place
static bool sphere_uses_ktable(int64_t batch, int64_t n1, int64_t n2, int dim, double beta, int flags) {
#ifdef GABO_SPH_NO_KT      /* A/B: the round-3 epilogue everywhere */
    return false;
#else
    const bool scaled = (flags & GABO_OUT_MASK) == GABO_OUT_GAUSSIAN && beta > 1e-30 && beta < 1000.0;
    return scaled && beta <= kSphKtMaxBeta && dim <= 16 && n2 >= 1024 && (double)batch * (double)n1 * (double)n2 >= (double)(1 << 22);
#endif
}

__global__ __launch_bounds__(256) void sphere_ktable_kernel(double beta, double* __restrict__ table) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < kSphKtSlots) sphere_kt_build_row(s, SphKtBaseRow::load(s), beta, table);
}
}  // namespace gabo

extern "C" int64_t gabo_sphere_ktable_doubles(void) { return (int64_t)gabo::kSphKtSlots * gabo::kSphKtStride; }

extern "C" int gabo_sphere_pairwise_uses_ktable(int64_t batch, int64_t n1, int64_t n2, int dim, double beta, int flags, int diag) {
    return (!diag && gabo::sphere_uses_ktable(batch, n1, n2, dim, beta, flags)) ? 1 : 0;
}

extern "C" int gabo_sphere_ktable_build(double beta, double* table, gabo_stream_t stream) {
    if (!table || !(beta > 1e-30) || beta > gabo::kSphKtMaxBeta) return GABO_ERR_ARG;
    hipLaunchKernelGGL(gabo::sphere_ktable_kernel, dim3((gabo::kSphKtSlots + 255) / 256), dim3(256), 0, (hipStream_t)stream, beta, table);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

extern "C" int gabo_sphere_pairwise_cached(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2,
                                           int dim, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags,
                                           int diag, const double* ktable, gabo_stream_t stream);

extern "C" int gabo_sphere_pairwise(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2,
                                    int dim, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags,
                                    int diag, gabo_stream_t stream) {
    return gabo_sphere_pairwise_cached(x1, x2, out, batch, n1, n2, dim, x1_batch_stride, x2_batch_stride, beta, flags, diag, nullptr, stream);
}

extern "C" int gabo_sphere_pairwise_cached(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2,
                                           int dim, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags,
                                           int diag, const double* ktable, gabo_stream_t stream) {
    if (batch < 0 || n1 < 0 || n2 < 0 || x1_batch_stride < 0 || x2_batch_stride < 0) return GABO_ERR_ARG;
    if (dim < 1) return GABO_ERR_DIM;
    if (batch == 0 || n1 == 0 || n2 == 0) return GABO_OK;
    if (!x1 || !x2 || !out) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (diag) {
        if (n1 != n2) return GABO_ERR_ARG;
        int64_t tot = batch * n1;
        hipLaunchKernelGGL(gabo::sphere_diag_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x1, x2, out, batch, n1,
                           dim, x1_batch_stride, x2_batch_stride, beta, flags);
    } else {
        if (batch > 65535) return GABO_ERR_ARG;
        const int mode = flags & GABO_OUT_MASK;
        // the usual case: table-driven epilogues; outside it the global polynomial with the clamped exp
        const bool scaled = mode == GABO_OUT_GAUSSIAN && beta > 1e-30 && beta < 1000.0;
        // large Gram matrices with beta <= 4: the kernel-value table (sphere_gauss_finish_kt), blocks of 1024 threads that build it once each
        const bool kt = gabo::sphere_uses_ktable(batch, n1, n2, dim, beta, flags);
        // blocks of the kernel-value table: 1024 threads (one table slot per thread, one block per CU) while the grid is about one round of
        // the 256 CUs; beyond that two co-resident 512-thread blocks per CU, whose prologues (table build, no stores in flight) overlap each
        // other's loops (measured, 1024 / 512 threads: N = 4096: 31.1 / 33.0 us, 8192: 142 / 131, 16384: 525 / 470; round-3 epilogue: 36.4 / 141 / 516)
#ifndef GABO_SPH_KT_THREADS
#define GABO_SPH_KT_THREADS ((((n2 + 1023) / 1024) * ((n1 + 16 * gabo::kSphKtChunks - 1) / (16 * gabo::kSphKtChunks)) * batch <= 320) ? 1024 : 512)
#endif
        const int kt_threads = GABO_SPH_KT_THREADS;
        int threads = kt ? kt_threads : (n2 >= 256 ? 256 : (n2 > 128 ? 192 : (n2 > 64 ? 128 : 64)));
        int64_t col_blocks = (n2 + threads - 1) / threads;
        // 16-row chunks per block: as many as keep >= 1024 blocks in flight (each wave re-uses its x2 fragments across the chunks)
#ifndef GABO_SPH_CHUNKS
#define GABO_SPH_CHUNKS 4
#endif
        static_assert(GABO_SPH_CHUNKS <= gabo::kSphMaxChunks, "LDS copy of the x1 rows");
        int chunks = kt ? gabo::kSphKtChunks : GABO_SPH_CHUNKS;
        while (chunks > 1 && col_blocks * ((n1 + 16 * chunks - 1) / (16 * chunks)) * batch < (kt ? 256 * (1024 / kt_threads) : 1024)) chunks >>= 1;
        const int rows = 16 * chunks;
        int64_t row_chunks = (n1 + rows - 1) / rows;
        int64_t tiles_x = col_blocks * row_chunks;
        if (flags & GABO_SYMMETRIC) {
            if (n1 != n2) return GABO_ERR_ARG;
            tiles_x = 0;
            for (int64_t cg = 0; cg < col_blocks; ++cg) tiles_x += gabo::sym_chunks_of(cg, threads, rows, row_chunks);
        }
        if (tiles_x > 0x7fffffffLL) return GABO_ERR_ARG;
        gabo::SphPoly poly = {};
        if (scaled) {
            const double sc = 4.0 * beta, pi = 3.14159265358979311600e+00;
            for (int k = 0; k <= gabo::kSphWDeg; ++k) poly.w[k] = gabo::kSphWHost[k] * sc;
            poly.a = 2.0 * pi * sqrt(beta);
            poly.b = beta * (pi * pi);
        }
        // results larger than the L2 caches (8 x 4 MB) are written with streaming stores: nothing of them would survive in L2 for a
        // consumer anyway (N = 4096: 41.5 -> 40.3 us)
#ifndef GABO_SPH_NT_BYTES
#define GABO_SPH_NT_BYTES (32ll << 20)
#endif
        const bool streaming = batch * n1 * n2 * 8 > GABO_SPH_NT_BYTES && !(flags & GABO_SYMMETRIC);
#ifdef GABO_SPH_NO_PW      /* A/B: the round-2 epilogue (global degree-17 polynomial) for the usual range of beta too */
#define GABO_SPH_PW_OF(SC) false
#else
#define GABO_SPH_PW_OF(SC) SC
#endif
#define GABO_SPH_LAUNCH_NT(M, SC, K, NT_)                                                                                          \
    hipLaunchKernelGGL((gabo::sphere_pairwise_kernel<M, SC, K, NT_, GABO_SPH_PW_OF(SC)>), dim3((unsigned)tiles_x, (unsigned)batch),     \
                       dim3(threads), 0, st, x1, x2, out, n1, n2, dim, x1_batch_stride, x2_batch_stride, (int)col_blocks, (int)row_chunks, \
                       chunks, beta, flags, poly, (const double*)nullptr)
#define GABO_SPH_LAUNCH_KT(K, NT_)                                                                                                 \
    hipLaunchKernelGGL((gabo::sphere_pairwise_kernel<GABO_OUT_GAUSSIAN, true, K, NT_, false, true>), dim3((unsigned)tiles_x, (unsigned)batch), \
                       dim3(threads), 0, st, x1, x2, out, n1, n2, dim, x1_batch_stride, x2_batch_stride, (int)col_blocks, (int)row_chunks, \
                       chunks, beta, flags, poly, ktable)
    // (the kernel-value table makes the kernel store-bound, and under the resulting back-pressure streaming stores measured far slower than
    // plain ones: N = 4096: 43.9 us with `nt`, 31.6 without - tools/ab_sphere.py, round 4)
#ifndef GABO_SPH_KT_NT
#define GABO_SPH_KT_NT 0
#endif
#define GABO_SPH_LAUNCH_KT_KS(K)                                                                                                   \
    do {                                                                                                                           \
        if (streaming && GABO_SPH_KT_NT) GABO_SPH_LAUNCH_KT(K, (GABO_SPH_KT_NT != 0));                                             \
        else GABO_SPH_LAUNCH_KT(K, false);                                                                                         \
    } while (0)
#define GABO_SPH_LAUNCH_KS(M, SC, K)                                                                                               \
    do {                                                                                                                           \
        if (streaming) GABO_SPH_LAUNCH_NT(M, SC, K, true);                                                                         \
        else GABO_SPH_LAUNCH_NT(M, SC, K, false);                                                                                  \
    } while (0)
#define GABO_SPH_LAUNCH(M, SC)                                                                                                     \
    do {                                                                                                                           \
        switch (dim <= 16 ? (dim + 3) / 4 : 0) {                                                                                   \
            case 1: GABO_SPH_LAUNCH_KS(M, SC, 1); break;                                                                           \
            case 2: GABO_SPH_LAUNCH_KS(M, SC, 2); break;                                                                           \
            case 3: GABO_SPH_LAUNCH_KS(M, SC, 3); break;                                                                           \
            case 4: GABO_SPH_LAUNCH_KS(M, SC, 4); break;                                                                           \
            default: GABO_SPH_LAUNCH_KS(M, SC, 0); break;                                                                          \
        }                                                                                                                          \
    } while (0)
        if (kt) {
            switch ((dim + 3) / 4) {
                case 1: GABO_SPH_LAUNCH_KT_KS(1); break;
                case 2: GABO_SPH_LAUNCH_KT_KS(2); break;
                case 3: GABO_SPH_LAUNCH_KT_KS(3); break;
                default: GABO_SPH_LAUNCH_KT_KS(4); break;
            }
        } else if (mode == GABO_OUT_DISTANCE) GABO_SPH_LAUNCH(GABO_OUT_DISTANCE, false);
        else if (mode == GABO_OUT_LAPLACE) GABO_SPH_LAUNCH(GABO_OUT_LAPLACE, false);
        else if (scaled) GABO_SPH_LAUNCH(GABO_OUT_GAUSSIAN, true);
        else GABO_SPH_LAUNCH(GABO_OUT_GAUSSIAN, false);
#undef GABO_SPH_LAUNCH
#undef GABO_SPH_LAUNCH_KS
#undef GABO_SPH_LAUNCH_NT
#undef GABO_SPH_LAUNCH_KT_KS
#undef GABO_SPH_LAUNCH_KT
        if (flags & GABO_SYMMETRIC) {
            int tiles = (int)((n1 + 31) / 32);
            hipLaunchKernelGGL((gabo::mirror_upper_kernel<1>), dim3((unsigned)((int64_t)tiles * (tiles + 1) / 2), (unsigned)batch),
                               dim3(256), 0, st, out, n1, tiles);
        }
    }
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
