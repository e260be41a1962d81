// Random SPD matrices with the distribution of spd_sample (Riemannian_utils/spd_utils.py:290-306): eigenvalues U[min_eig, max_eig],
// eigenvectors = the orthogonal factor of a Gaussian matrix, X = Q diag(lambda) Q^T - drawn on the device, one lane per matrix.
// The reference draws raw_samples of them per BO iteration through manifold.rand on the host (manifold_optimize.py:288); this is
// the opt-in device path of that stage (same distribution, its own counter-based random stream: Philox4x32-10 keyed by the seed,
// counter = (matrix index, draw index), so a sample does not depend on the launch geometry).
#include "gabo_device.hpp"
#include "gabo_philox.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

// lane-private d x d scratch in LDS: element e of lane l at q[e * 64 + l] (bank-conflict free, dynamically indexable)
// out_stride: doubles between consecutive samples in `out` (the sweep driver writes them into rows [value, Mandel vector]: spd_sweep.hip)
__global__ __launch_bounds__(64) void spd_sample_kernel(double* __restrict__ out, int64_t first, int64_t n, int d, double min_eig,
                                                        double max_eig, uint64_t seed, int mandel, int64_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    if (i >= n) return;
    double* q = lds + lane;
    Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint64_t)(first + i), 0u};   // counter = GLOBAL sample index
    const int dd = d * d;
    for (int e = 0; e < dd; e += 2) {
        double z0, z1;
        rng.normal2(z0, z1);
        q[e * 64] = z0;
        if (e + 1 < dd) q[(e + 1) * 64] = z1;
    }
    // orthonormalise the columns: modified Gram-Schmidt, two passes per column (orthogonal to rounding for any conditioning met here)
    for (int c = 0; c < d; ++c) {
        for (int pass = 0; pass < 2; ++pass) {
            for (int p = 0; p < c; ++p) {
                double dot = 0.0;
                for (int r = 0; r < d; ++r) dot = __builtin_fma(q[(r * d + p) * 64], q[(r * d + c) * 64], dot);
                for (int r = 0; r < d; ++r) q[(r * d + c) * 64] = __builtin_fma(-dot, q[(r * d + p) * 64], q[(r * d + c) * 64]);
            }
        }
        double nn = 0.0;
        for (int r = 0; r < d; ++r) nn = __builtin_fma(q[(r * d + c) * 64], q[(r * d + c) * 64], nn);
        const double inv = 1.0 / __builtin_sqrt(nn);
        for (int r = 0; r < d; ++r) q[(r * d + c) * 64] *= inv;
    }
    // scale column k by sqrt(lambda_k):  X = (Q sqrt(L)) (Q sqrt(L))^T is symmetric positive definite by construction
    for (int k = 0; k < d; k += 2) {
        double u1, u2;
        rng.uniform2(u1, u2);
        const double s0 = __builtin_sqrt(min_eig + (max_eig - min_eig) * u2);
        const double s1 = __builtin_sqrt(min_eig + (max_eig - min_eig) * (1.0 - u1));
        for (int r = 0; r < d; ++r) {
            q[(r * d + k) * 64] *= s0;
            if (k + 1 < d) q[(r * d + k + 1) * 64] *= s1;
        }
    }
    if (mandel) {
        double* o = out + i * out_stride;
        for (int r = 0; r < d; ++r)
            for (int c = 0; c <= r; ++c) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s = __builtin_fma(q[(r * d + k) * 64], q[(c * d + k) * 64], s);
                o[mandel_pos(d, r, c)] = (r == c) ? s : s * kSqrt2;
            }
    } else {
        double* o = out + i * out_stride;
        for (int r = 0; r < d; ++r)
            for (int c = 0; c <= r; ++c) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s = __builtin_fma(q[(r * d + k) * 64], q[(c * d + k) * 64], s);
                o[r * d + c] = s;
                o[c * d + r] = s;
            }
    }
}


// The same sampler for d <= 8 with the matrix in the lane's REGISTERS (compile-time indices): the draws, their order and every arithmetic statement
// are those of spd_sample_kernel above - same stream, same bits - without the LDS round trip of each of its ~10 d^3 accesses (the kernel is one
// dependent chain per lane: 26 us for 256 matrices of order 5 under rocprofv3, on the critical path of an acquisition sweep between its set-up and
// its scoring launch).
template <int D>
__global__ __launch_bounds__(64) void spd_sample_reg_kernel(double* __restrict__ out, int64_t first, int64_t n, double min_eig, double max_eig,
                                                            uint64_t seed, int mandel, int64_t out_stride) {
    constexpr int dd = D * D;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double q[dd + 1];
    Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint64_t)(first + i), 0u};
    static_for<(dd + 1) / 2>([&](auto hh) {
        constexpr int e = 2 * decltype(hh)::value;
        double z0, z1;
        rng.normal2(z0, z1);
        q[e] = z0;
        q[e + 1] = z1;          // (e + 1 == dd for odd dd: the spare slot)
    });
    static_for<D>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        static_for<2>([&](auto) {
            static_for<c>([&](auto pp) {
                constexpr int p = decltype(pp)::value;
                double dot = 0.0;
                static_for<D>([&](auto rr) { constexpr int r = decltype(rr)::value; dot = __builtin_fma(q[r * D + p], q[r * D + c], dot); });
                static_for<D>([&](auto rr) { constexpr int r = decltype(rr)::value; q[r * D + c] = __builtin_fma(-dot, q[r * D + p], q[r * D + c]); });
            });
        });
        double nn = 0.0;
        static_for<D>([&](auto rr) { constexpr int r = decltype(rr)::value; nn = __builtin_fma(q[r * D + c], q[r * D + c], nn); });
        const double inv = 1.0 / __builtin_sqrt(nn);
        static_for<D>([&](auto rr) { constexpr int r = decltype(rr)::value; q[r * D + c] *= inv; });
    });
    static_for<(D + 1) / 2>([&](auto hh) {
        constexpr int k = 2 * decltype(hh)::value;
        double u1, u2;
        rng.uniform2(u1, u2);
        const double s0 = __builtin_sqrt(min_eig + (max_eig - min_eig) * u2);
        const double s1 = __builtin_sqrt(min_eig + (max_eig - min_eig) * (1.0 - u1));
        static_for<D>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            q[r * D + k] *= s0;
            if constexpr (k + 1 < D) q[r * D + k + 1] *= s1;
        });
    });
    double* o = out + i * out_stride;
    static_for<D>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        static_for<r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            double sacc = 0.0;
            static_for<D>([&](auto kk) { constexpr int k = decltype(kk)::value; sacc = __builtin_fma(q[r * D + k], q[c * D + k], sacc); });
            if (mandel) {
                o[mandel_pos(D, r, c)] = (r == c) ? sacc : sacc * kSqrt2;
            } else {
                o[r * D + c] = sacc;
                o[c * D + r] = sacc;
            }
        });
    });
}

// launches the register form for d <= 8, the LDS form above it
static int launch_spd_sample(double* out, int64_t out_stride, int64_t first, int64_t n, int d, double min_eig, double max_eig, uint64_t seed, int mandel,
                             hipStream_t st) {
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
#define GABO_CASE(DD)                                                                                                              \
    case DD:                                                                                                                       \
        hipLaunchKernelGGL((spd_sample_reg_kernel<DD>), grid, block, 0, st, out, first, n, min_eig, max_eig, seed, mandel, out_stride); \
        break;
    switch (d) {
        GABO_CASE(1) GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8)
        default: {
            size_t lds = (size_t)d * d * 64 * sizeof(double);
            hipLaunchKernelGGL(spd_sample_kernel, grid, block, lds, st, out, first, n, d, min_eig, max_eig, seed, mandel, out_stride);
        }
    }
#undef GABO_CASE
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
}  // namespace gabo

namespace gabo {
// samples first ... first + n - 1 as Mandel vectors at out + i * out_stride (arguments checked by the caller)
int spd_sample_rows(double* out, int64_t out_stride, int64_t first, int64_t n, int d, double min_eig, double max_eig, uint64_t seed, hipStream_t st) {
    return launch_spd_sample(out, out_stride, first, n, d, min_eig, max_eig, seed, 1, st);
}
}  // namespace gabo

extern "C" int gabo_spd_sample_range(double* out, int64_t first, int64_t n, int d, double min_eig, double max_eig, uint64_t seed,
                                     int mandel, gabo_stream_t stream) {
    if (d < 1 || d > 16) return GABO_ERR_DIM;
    if (first < 0 || n < 0 || !(min_eig > 0.0) || !(max_eig >= min_eig)) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    if (!out) return GABO_ERR_ARG;
    return gabo::launch_spd_sample(out, (int64_t)(mandel ? d * (d + 1) / 2 : d * d), first, n, d, min_eig, max_eig, seed, mandel, (hipStream_t)stream);
}

extern "C" int gabo_spd_sample(double* out, int64_t n, int d, double min_eig, double max_eig, uint64_t seed, int mandel,
                               gabo_stream_t stream) {
    return gabo_spd_sample_range(out, 0, n, d, min_eig, max_eig, seed, mandel, stream);
}
