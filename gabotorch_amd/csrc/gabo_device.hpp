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
// 1/x: v_rcp_f64 seed + two Newton steps (no denormal/overflow scaling: operands here are O(1e+-150) at worst).
__device__ __forceinline__ double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// sqrt(x), x >= 0: v_rsq_f64 seed, two coupled Goldschmidt steps and a final residual correction.  sqrt(0) = 0.
__device__ __forceinline__ double sqrt_pos(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double dres = __builtin_fma(-g, g, x);
    g = __builtin_fma(dres, h, g);
    return x == 0.0 ? 0.0 : g;
}

// 1/sqrt(x), x > 0
__device__ __forceinline__ double rsqrt_pos(double x) {
    double y = __builtin_amdgcn_rsq(x);
    // Newton: y <- y * (1.5 - 0.5 x y^2), twice
    double hx = 0.5 * x;
    double t = __builtin_fma(-hx * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    t = __builtin_fma(-hx * y, y, 0.5);
    return __builtin_fma(y, t, y);
}

__device__ __forceinline__ double copysign_d(double mag, double sgn) { return __builtin_copysign(mag, sgn); }

}  // namespace gabo
