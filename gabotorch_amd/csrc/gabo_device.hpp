// Device-side building blocks shared by the gfx950 kernels: compile-time loops, packed-triangle indexing and the
// fp64 scalar math (reciprocal / sqrt by hardware seed + Newton steps) the per-lane eigensolver is built from.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

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

}  // namespace gabo
