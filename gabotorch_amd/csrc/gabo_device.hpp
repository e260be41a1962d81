// Device-side building blocks shared by the gfx950 kernels: compile-time loops, packed-triangle indexing and the
// fp64 scalar math (reciprocal / sqrt by hardware seed + Newton steps) the per-lane eigensolver is built from.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

// Development instrumentation (builds with -DGABO_TR_CLOCKS only; tools/tr_clocks.py): block 0 / lane 0 appends (tag, s_memtime) pairs
// to a per-translation-unit buffer that the exported gabo_debug_clocks of that unit copies out.  Compiles to nothing otherwise.
#ifdef GABO_TICK
/* (the translation unit brought its own recorder: spd_tr_solve_duo.hip with -DGABO_DUO_TIMES) */
#elif defined(GABO_TR_CLOCKS)
#ifndef GABO_TR_CLOCKS_BLOCK
#define GABO_TR_CLOCKS_BLOCK 0      /* the restart (block) whose waves record: -DGABO_TR_CLOCKS_BLOCK=<index of a restart that runs to maxiter> */
#endif
static __device__ long long gabo_clk_buf[8192];
static __device__ int gabo_clk_n;
#define GABO_TICK(tag)                                                          \
    do {                                                                        \
        if (threadIdx.x == 0 && blockIdx.x == GABO_TR_CLOCKS_BLOCK) {           \
            int k_ = gabo_clk_n++;                                              \
            if (k_ < 4096) {                                                    \
                gabo_clk_buf[2 * k_] = (tag);                                   \
                gabo_clk_buf[2 * k_ + 1] = (long long)__builtin_amdgcn_s_memtime(); \
            }                                                                   \
        }                                                                       \
    } while (0)
#else
#define GABO_TICK(tag) do { } while (0)
#endif

namespace gabo {

// ---- compile-time loops: every index reaching a register array is a constant, so nothing lands in scratch -----
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
// f(ic<0>), f(ic<1>), ... f(ic<N-1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
// f(ic<Hi>), f(ic<Hi-1>), ... f(ic<Lo>)   (inclusive, descending)
template <int Hi, int Lo, class F>
__device__ __forceinline__ void static_for_down(F&& f) {
    if constexpr (Hi >= Lo) {
        static_for<Hi - Lo + 1>([&](auto k) { f(std::integral_constant<int, Hi - decltype(k)::value>{}); });
    }
}

// packed lower triangle, row-major: (r, c<=r) -> r(r+1)/2 + c
__host__ __device__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }
__host__ __device__ constexpr int tri_size(int d) { return d * (d + 1) / 2; }

// Mandel position of matrix entry (r, c), r >= c: the main diagonal first, then super-diagonal 1, 2, ...
// (reference layout: Riemannian_utils/spd_utils_torch.py:183-187).
__host__ __device__ constexpr int mandel_pos(int d, int r, int c) {
    int k = r - c;                       // which off-diagonal
    return k * d - k * (k - 1) / 2 + c;  // entries before diagonal k: sum_{t<k}(d-t), then position c along it
}

constexpr double kInvSqrt2 = 0.70710678118654752440;
constexpr double kSqrt2 = 1.41421356237309504880;

// ---- fp64 scalar math -------------------------------------------------------------------------------------------
// Measured on gfx950 (tools/ubench.hip): v_rcp_f64 and v_rsq_f64 return seeds good to 2^-24.  No denormal/overflow
// scaling anywhere below: operands on this path are O(1e+-100) at worst.

// 1/x.  e = 1 - x r0 (|e| <= 2^-24);  r0 (1 + e + e^2) has error e^3 = 2^-72: one seed + 3 FMAs, ~1 ulp.
__device__ __forceinline__ double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    double e2 = __builtin_fma(e, e, e);
    return __builtin_fma(r, e2, r);
}

// 1/x to ~2^-47 (one Newton step): for quantities that only need to be right to a few 1e-15
__device__ __forceinline__ double rcp_nr1(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// sqrt(x), x > 0: seed, one coupled Goldschmidt step (2^-47), then the residual correction g += (x - g^2) h.
__device__ __forceinline__ double sqrt_nz(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double dres = __builtin_fma(-g, g, x);
    return __builtin_fma(dres, h, g);
}

// sqrt(x), x > 0, from the bare seed by ONE cubic correction: g0 = x y0, r = 1 - g0 y0 (|r| <= 2^-22), sqrt x = g0 (1 - r)^(-1/2) =
// g0 (1 + r/2 + 3 r^2/8) + O(r^3): five instructions after the seed where the Goldschmidt form above takes seven; the rounding of g0 enters
// r through the same FMA that uses it, so what is left is half an ulp of g0 and the final rounding.
__device__ __forceinline__ double sqrt_nz_cubic(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double r = __builtin_fma(-g, y, 1.0);
    double p = __builtin_fma(0.375, r, 0.5) * r;
    return __builtin_fma(g, p, g);
}

// Sum over the 64 lanes of a wave, the same bits in every lane.  __shfl_xor on a double is two ds_bpermute_b32 per step: six dependent
// round trips through the LDS crossbar (~100 cycles each) in kernels whose every phase waits for such a sum.  Here the four steps inside a
// row of 16 lanes are DPP moves on the vector pipe (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: each pairs lanes symmetrically, so
// all 16 lanes of a row end with the same bits), the rows are combined with row_bcast15 / row_bcast31 (rows 1 and 3 += their lower
// neighbour, then row 3 += row 1: lane 63 holds (r3 + r2) + (r1 + r0)) and lane 63 is broadcast through two v_readlane: ~21 instructions,
// no LDS.  All 64 lanes must be active.  GABO_WAVE_SUM_SHFL: the butterfly (A/B).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_allsum(double v) {
#ifdef GABO_WAVE_SUM_SHFL
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
#else
    v += dpp_fetch<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_fetch<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_fetch<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_fetch<0x140, 0xf>(v);       // row_mirror
    v += dpp_fetch<0x142, 0xa>(v);       // row_bcast15 into rows 1 and 3 (the other rows add the `old` operand: +0.0)
    v += dpp_fetch<0x143, 0xc>(v);       // row_bcast31 into rows 2 and 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
#endif
}

// 1/sqrt(x), x > 0: e = 1 - x y0^2 (|e| <= 2^-23); y0 (1 + e/2 + 3e^2/8) leaves an e^3 error.
__device__ __forceinline__ double rsqrt_nz(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    double p = e * __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y, p, y);
}

// sqrt(x), x >= 0 (sqrt(0) = 0)
__device__ __forceinline__ double sqrt_pos(double x) {
    double g = sqrt_nz(x);
    return x == 0.0 ? 0.0 : g;
}

__device__ __forceinline__ double copysign_d(double mag, double sgn) { return __builtin_copysign(mag, sgn); }

// a * b + c with the addend read from an SGPR pair (c wave-uniform).  hipcc would pick the two-address v_fmac_f64 and
// first copy the coefficient into the accumulator with two v_mov_b32; VOP3 v_fma_f64 takes the scalar addend directly.
__device__ __forceinline__ double fma_s(double a, double b, double c_uniform) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}

// Polynomial coefficients of acos_fast / exp_neg.  fp64 literals cannot be VALU inline operands and hipcc re-materialises
// each one with two v_mov_b32 in front of every v_fmac (tripling the VALU cost of a Horner step).  A kernel therefore
// loads the table ONCE into registers (`MathRegs::load()`, laundered through an empty asm so the values are opaque and
// cannot be re-materialised) and every Horner step is one v_fma_f64 with a live register addend.
struct MathRegs {
    double pS[6], qS[4];           // fdlibm e_acos.c: R(z) = z P(z) / Q(z)
    double pio2_hi, pio2_lo, pi;
    double log2e, ln2_hi, ln2_lo;
    double inv_fact[12];           // 1/13!, 1/12!, ..., 1/2!

#ifdef GABO_MATH_SGPR   /* coefficients pinned in SGPRs (kernels with spare scalar registers): frees 56 VGPRs */
    __device__ __forceinline__ static double pin(double v) {
        asm volatile("" : "+s"(v));
        return v;
    }
    __device__ __forceinline__ static double fmac(double a, double b, double c) { return fma_s(a, b, c); }
#else
    __device__ __forceinline__ static double pin(double v) {
        asm volatile("" : "+v"(v));
        return v;
    }
    __device__ __forceinline__ static double fmac(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
    __device__ __forceinline__ static MathRegs load() {
        MathRegs t;
        const double pS[6] = {1.66666666666666657415e-01, -3.25565818622400915405e-01, 2.01212532134862925881e-01,
                              -4.00555345006794114027e-02, 7.91534994289814532176e-04, 3.47933107596021167570e-05};
        const double qS[4] = {-2.40339491173441421878e+00, 2.02094576023350569471e+00, -6.88283971605453293030e-01,
                              7.70381505559019352791e-02};
        const double f[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
                              1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};
        static_for<6>([&](auto i) { t.pS[decltype(i)::value] = pin(pS[decltype(i)::value]); });
        static_for<4>([&](auto i) { t.qS[decltype(i)::value] = pin(qS[decltype(i)::value]); });
        static_for<12>([&](auto i) { t.inv_fact[decltype(i)::value] = pin(f[decltype(i)::value]); });
        t.pio2_hi = pin(1.57079632679489655800e+00);
        t.pio2_lo = pin(6.12323399573676603587e-17);
        t.pi = pin(3.14159265358979311600e+00);
        t.log2e = pin(1.44269504088896338700e+00);
        t.ln2_hi = pin(6.93147180369123816490e-01);
        t.ln2_lo = pin(1.90821492927058770002e-10);
        return t;
    }
};

// acos(c) for |c| < 1 (callers clamp to [-1+1e-15, 1-1e-15]), branch-free.  Same rational approximation as fdlibm's
// e_acos.c: R(z) = z P(z)/Q(z);  |c| <= 1/2: pi/2 - (c + c R(c^2));  c > 1/2: 2 (s + s R(z)), z = (1-c)/2, s = sqrt z;
// c < -1/2: pi - 2 (s + s R(z)), z = (1+c)/2.  Both ranges share one evaluation of R on a selected z.  1 ulp measured.
__device__ __forceinline__ double acos_fast(double c, const MathRegs& t) {
    double a = __builtin_fabs(c);
    bool small = a <= 0.5;
    double z = small ? a * a : 0.5 * (1.0 - a);
    double p = MathRegs::fmac(z * t.pS[5], 1.0, t.pS[4]);
    p = MathRegs::fmac(z, p, t.pS[3]);
    p = MathRegs::fmac(z, p, t.pS[2]);
    p = MathRegs::fmac(z, p, t.pS[1]);
    p = MathRegs::fmac(z, p, t.pS[0]);
    p = p * z;
    double q = MathRegs::fmac(z * t.qS[3], 1.0, t.qS[2]);
    q = MathRegs::fmac(z, q, t.qS[1]);
    q = MathRegs::fmac(z, q, t.qS[0]);
    q = __builtin_fma(z, q, 1.0);
    double r = p * rcp(q);
    double s = sqrt_nz(small ? 1.0 : z);
    double res_small = t.pio2_hi - (c - __builtin_fma(-c, r, t.pio2_lo));   // pio2_hi - (c - (pio2_lo - c r))
    double w = __builtin_fma(s, r, s);                                        // s + s r
    double res_large = (c > 0.0) ? 2.0 * w : __builtin_fma(-2.0, w - t.pio2_lo, t.pi);
    return small ? res_small : res_large;
}

// exp(x) for x <= 0 (kernel values): k = rint(x log2 e), r = x - k ln2 (two-word ln2), degree-13 Taylor polynomial in r
// (|r| <= 0.3466: truncation 4e-18), scaled by 2^k with v_ldexp_f64 (which also handles gradual underflow to 0).  1 ulp.
__device__ __forceinline__ double exp_neg(double x, const MathRegs& t) {
    x = x < -800.0 ? -800.0 : x;
    double k = __builtin_rint(x * t.log2e);
    double r = __builtin_fma(-k, t.ln2_lo, __builtin_fma(-k, t.ln2_hi, x));
    double p = MathRegs::fmac(r * t.inv_fact[0], 1.0, t.inv_fact[1]);
    p = MathRegs::fmac(p, r, t.inv_fact[2]);
    p = MathRegs::fmac(p, r, t.inv_fact[3]);
    p = MathRegs::fmac(p, r, t.inv_fact[4]);
    p = MathRegs::fmac(p, r, t.inv_fact[5]);
    p = MathRegs::fmac(p, r, t.inv_fact[6]);
    p = MathRegs::fmac(p, r, t.inv_fact[7]);
    p = MathRegs::fmac(p, r, t.inv_fact[8]);
    p = MathRegs::fmac(p, r, t.inv_fact[9]);
    p = MathRegs::fmac(p, r, t.inv_fact[10]);
    p = MathRegs::fmac(p, r, t.inv_fact[11]);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// max(x, bound) for x known not to be a signalling NaN (hipcc otherwise canonicalises x with an extra v_max_f64 x, x first)
__device__ __forceinline__ double max_raw(double x, double bound_uniform) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(bound_uniform));
    return r;
}

// Table-assisted exp(x) for x <= 0 (kernel values of the write-bound Gram kernels): x = (64 e + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^e * 2^(j/64) * (1 + r + ... + r^5/120) (3.5e-17 truncation).  17 VALU instructions and one LDS read against ~37 for OCML's
// exp (whose Horner steps each re-materialise a literal).  `tab`: the 64 entries of kExp2Tab copied to LDS by the kernel; `c`: kExpTabC
// held in SGPRs ([0] ln2/64 high part with 33 bits so that k * hi is exact, [1] low part, [2] 64/ln2, [3..5] 1/120, 1/24, 1/6);
// `c3_vgpr`: c[3] in a VGPR (a VALU instruction reads one scalar operand).
static __constant__ double kExpTabC[6] = {0.010830424695086549, 1.162596423439437e-12, 92.33248261689366, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0};
static __constant__ double kExp2Tab[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284, 1.0442737824274138, 1.0556451783605572, 1.0671404006768237,
    1.0787607977571199, 1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418, 1.1387886347566916,
    1.1511892299529827, 1.1637248587775775, 1.1763969916502812, 1.189207115002721, 1.202156731452703, 1.215247359980469,
    1.22848053610687, 1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783, 1.2968395546510096,
    1.3109612115247644, 1.3252366431597413, 1.339667524053303, 1.3542555469368927, 1.3690024229745905, 1.383909881963832,
    1.3989796725383112, 1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647, 1.4768261459394993,
    1.4929077282912648, 1.5091644275934228, 1.5255981507445384, 1.5422108254079407, 1.559004400237837, 1.5759808451078865,
    1.593142151342267, 1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364, 1.681792830507429,
    1.7001063537185235, 1.718619298122478, 1.7373338352737062, 1.7562521603732995, 1.7753764925265212, 1.7947090750031072,
    1.8142521755003989, 1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656, 1.9152065613971474,
    1.9360617934922943, 1.9571441241754002, 1.978456026387951};

// CLAMP = false: for callers whose argument is bounded below by construction, |x| < 1e4 (k ln2/64 is exact for |k| < 2^20, so the reduced
// argument stays small and the result underflows cleanly through v_ldexp_f64 below -745).
template <bool CLAMP = true, class C>
__device__ __forceinline__ double exp_neg_tab(double x, const C& c, double c3_vgpr, const double* __restrict__ tab) {
    if constexpr (CLAMP) x = max_raw(x, -800.0);
    double k = __builtin_rint(x * c[2]);
    double r = __builtin_fma(-k, c[1], __builtin_fma(-k, c[0], x));
    double p = __builtin_fma(r, c3_vgpr, c[4]);
    p = __builtin_fma(p, r, c[5]);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;                                                 // exp(r) - 1
    int ki = (int)k;
    double t = tab[ki & 63];
    return __builtin_ldexp(__builtin_fma(t, p, t), ki >> 6);   // (v_ldexp_f64 also does the gradual underflow)
}

// The same with a 256-entry table (2 KB of LDS): |r| <= ln2/512, so exp(r) - 1 = r (1 + r/2 + r^2/6 + r^3/24) leaves r^5/120 = 3.8e-17 - one FMA
// less per value.  c: [0] ln2/256 head (k c[0] exact for |k| < 2^19), [1] tail, [2] 256/ln2, [3] 1/6; `c24_vgpr`: 1/24 in a VGPR.
// tools/sim/gen_exp_table.py generates the table and models the routine (2.2e-16 on [-40, 0]).
template <bool CLAMP = true, class C>
__device__ __forceinline__ double exp_neg_tab256(double x, const C& c, double c24_vgpr, const double* __restrict__ tab) {
    if constexpr (CLAMP) x = max_raw(x, -800.0);
    double k = __builtin_rint(x * c[2]);
    double r = __builtin_fma(-k, c[1], __builtin_fma(-k, c[0], x));
    double p = __builtin_fma(r, c24_vgpr, c[3]);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;                                                 // exp(r) - 1
    int ki = (int)k;
    double t = tab[ki & 255];
    return __builtin_ldexp(__builtin_fma(t, p, t), ki >> 8);
}

// exp(-y) for y >= 0 given as y (the caller's quantity is naturally positive: beta d^2): the same routine with the sign folded into the
// constants - c[2] = -256/ln2, and -y enters the reduction as a negated FMA addend - so that no instruction is spent on the negation.
template <bool CLAMP = true, class C>
__device__ __forceinline__ double exp_of_minus_tab256(double y, const C& c, double c24_vgpr, const double* __restrict__ tab) {
    if constexpr (CLAMP) y = __builtin_fmin(y, 800.0);
    double k = __builtin_rint(y * c[2]);                       // c[2] < 0: k = rint(-y 256 / ln2)
    double r = __builtin_fma(-k, c[1], __builtin_fma(-k, c[0], -y));
    double p = __builtin_fma(r, c24_vgpr, c[3]);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    int ki = (int)k;
    double t = tab[ki & 255];
    return __builtin_ldexp(__builtin_fma(t, p, t), ki >> 8);
}

// The same with the round-to-integer done by the FMA that scales the argument: km = y c[2] + 1.5 2^52 holds k = rint(y c[2]) in its low
// mantissa bits (two's complement in the low word for |k| < 2^31), k = km - 1.5 2^52 is exact: v_fma + v_add replace v_mul + v_rndne + v_cvt_i32.
// `magic_vgpr`: 1.5 2^52 in a VGPR (a VALU instruction reads one scalar operand, and c[2] is it).  |y| < 1e4 by construction: no clamp.
template <class C>
__device__ __forceinline__ double exp_of_minus_tab256_magic(double y, const C& c, double c24_vgpr, double magic_vgpr,
                                                            const double* __restrict__ tab) {
    double km = __builtin_fma(y, c[2], magic_vgpr);
    double k = km - magic_vgpr;
    double r = __builtin_fma(-k, c[1], __builtin_fma(-k, c[0], -y));
    double p = __builtin_fma(r, c24_vgpr, c[3]);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    int ki = __double2loint(km);
    double t = tab[ki & 255];
    return __builtin_ldexp(__builtin_fma(t, p, t), ki >> 8);
}

// log(x) for positive normal x (eigenvalues of an SPD matrix), fdlibm e_log.c scheme: x = 2^k (1+f), sqrt(1/2) <= 1+f < sqrt 2,
// s = f/(2+f), log(1+f) = f - hfsq + s (hfsq + R(s^2)), R a degree-7 minimax polynomial split into even and odd halves.
// OCML's log is 98 VALU instructions (double-double arithmetic); this is ~35 at 1 ulp.  Coefficients pinned in registers.
struct LogRegs {
    double lg[7], ln2_hi, ln2_lo, sqrt_half;
    __device__ __forceinline__ static LogRegs load() {
        LogRegs t;
        const double c[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
                             1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};
        static_for<7>([&](auto i) { t.lg[decltype(i)::value] = MathRegs::pin(c[decltype(i)::value]); });
        t.ln2_hi = MathRegs::pin(6.93147180369123816490e-01);
        t.ln2_lo = MathRegs::pin(1.90821492927058770002e-10);
        t.sqrt_half = MathRegs::pin(0.70710678118654752440);
        return t;
    }
};

__device__ __forceinline__ double log_pos(double x, const LogRegs& t) {
    double m = __builtin_amdgcn_frexp_mant(x);        // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    bool lo = m < t.sqrt_half;
    m = lo ? m + m : m;                               // [sqrt(1/2), sqrt 2)
    k = lo ? k - 1 : k;
    double f = m - 1.0;
    double s = f * rcp(2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * MathRegs::fmac(w, MathRegs::fmac(w, t.lg[5], t.lg[3]), t.lg[1]);
    double t2 = z * MathRegs::fmac(w, MathRegs::fmac(w, MathRegs::fmac(w, t.lg[6], t.lg[4]), t.lg[2]), t.lg[0]);
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return __builtin_fma(dk, t.ln2_hi, -((hfsq - __builtin_fma(s, hfsq + R, dk * t.ln2_lo)) - f));
}


// Table-assisted log(x) for positive normal x: k = exponent of x sqrt2, m = x 2^-k in [sqrt(1/2), sqrt 2), (c, l) = kLogTab[(exponent LSB, top 7
// mantissa bits) of m] (gabo_log_tab.hpp: c ~ 1 / bucket centre, l = -log c; both buckets next to 1 hold (1, 0) so that log 1 = 0 exactly and the
// logs of near-identity eigenvalues keep their relative accuracy), r = m c - 1 exactly (one FMA; |r| <= 2^-8, < 2^-7 next to 1),
// log x = k ln2 + l + log1p(r) with the degree-7 series of log1p.  <= 2 ulp (tools/sim/gen_log_table.py models it in numpy); 20 VALU
// instructions and one 16-byte LDS read, no reciprocal and no compare, against 35 with a quarter-rate v_rcp_f64 for log_pos.
// `tab`: the 256 (c, l) pairs copied to LDS by the kernel.
struct LogTabRegs {
    double sqrt2, c7, c6, c5, c4, c3, ln2_hi, ln2_lo;
    __device__ __forceinline__ static LogTabRegs load() {
        LogTabRegs t;
        t.sqrt2 = MathRegs::pin(1.41421356237309504880);
        t.c7 = MathRegs::pin(1.0 / 7.0);
        t.c6 = MathRegs::pin(-1.0 / 6.0);
        t.c5 = MathRegs::pin(0.2);
        t.c4 = MathRegs::pin(-0.25);
        t.c3 = MathRegs::pin(1.0 / 3.0);
        t.ln2_hi = MathRegs::pin(6.93147180369123816490e-01);
        t.ln2_lo = MathRegs::pin(1.90821492927058770002e-10);
        return t;
    }
};

__device__ __forceinline__ double log_tab(double x, const LogTabRegs& t, const double* __restrict__ tab) {
    const int ke = __builtin_amdgcn_frexp_exp(x * t.sqrt2);        // x sqrt2 = mant 2^ke, mant in [1/2, 1)
    const double m = __builtin_ldexp(x, 1 - ke);                   // [sqrt(1/2), sqrt 2) (to a rounding of the product above: the table covers [1/2, 2))
    const unsigned hi = (unsigned)__double2hiint(m);
    const double2 cl = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + ((hi >> 9) & 0xff0u));
    const double r = __builtin_fma(m, cl.x, -1.0);
    double p = MathRegs::fmac(r, t.c7, t.c6);
    p = MathRegs::fmac(p, r, t.c5);
    p = MathRegs::fmac(p, r, t.c4);
    p = MathRegs::fmac(p, r, t.c3);
    p = __builtin_fma(p, r, -0.5);
    const double lp = __builtin_fma(r * r, p, r);
    const double dk = (double)(ke - 1);
    return __builtin_fma(dk, t.ln2_hi, cl.y) + __builtin_fma(dk, t.ln2_lo, lp);
}

}  // namespace gabo
