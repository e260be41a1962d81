// Sphere pairwise kernel matrix on gfx950:  out_ij = f(acos(clamp(<x1_i, x2_j>, -1+1e-15, 1-1e-15)))
// replaces kernel_utils/kernels_sphere.py:71-94,118-134 + Riemannian_utils/sphere_utils_torch.py:12-55.
//
// The reference materialises two N1 x N2 x dim tensors and a degenerate bmm; here a block owns a ROWS x 256 output
// tile.  Lane = column j (so the fp64 stores of a wave are one contiguous 512 B run), each lane keeps ROWS running
// inner products in registers.  The x1 rows of the tile are wave-uniform and come through the scalar cache; the x2
// row of the lane is read once per tile.  HBM traffic = the output matrix (8 B per pair); this kernel is bound by the
// fp64 acos+exp epilogue and the output write, not by operand reads.
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

constexpr int kSphereRows = 8;

__device__ __forceinline__ double sphere_finish(double ip, double beta, int mode) {
    const double lo = -1.0 + 1e-15, hi = 1.0 - 1e-15;  // sphere_utils_torch.py:53
    double c = ip < lo ? lo : (ip > hi ? hi : ip);
    double dist = acos(c);                                // :55
    if (mode == GABO_OUT_DISTANCE) return dist;
    if (mode == GABO_OUT_LAPLACE) return exp(-(dist * beta));
    return exp(-((dist * dist) * beta));                  // kernels_sphere.py:91-93
}

// 1-D grid: block id -> (batch, row chunk, column group), column group fastest
__global__ __launch_bounds__(256) void sphere_pairwise_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                              double* __restrict__ out, int64_t n1, int64_t n2, int dim,
                                                              int64_t s1, int64_t s2, int col_blocks, int row_chunks,
                                                              double beta, int flags) {
    const int mode = flags & GABO_OUT_MASK;
    const int64_t bid = blockIdx.x;
    const int64_t cg = bid % col_blocks;
    const int64_t rc = (bid / col_blocks) % row_chunks;
    const int64_t b = bid / ((int64_t)col_blocks * row_chunks);
    const int64_t j = cg * blockDim.x + threadIdx.x;
    const int64_t jc = j < n2 ? j : n2 - 1;
    const int64_t i0 = rc * kSphereRows;
    const double* a = x1 + b * s1;                 // rows i0.. (uniform)
    const double* bj = x2 + b * s2 + jc * dim;     // this lane's point
    double acc[kSphereRows];
    static_for<kSphereRows>([&](auto r) { acc[decltype(r)::value] = 0.0; });
    for (int k = 0; k < dim; ++k) {
        double y = bj[k];
        static_for<kSphereRows>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            int64_t i = i0 + r < n1 ? i0 + r : n1 - 1;
            acc[r] = __builtin_fma(a[i * dim + k], y, acc[r]);
        });
    }
    if (j < n2) {
        double* o = out + b * n1 * n2 + j;
        static_for<kSphereRows>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            if (i0 + r < n1) o[(i0 + r) * n2] = sphere_finish(acc[r], beta, mode);
        });
    }
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
    out[g] = sphere_finish(acc, beta, flags & GABO_OUT_MASK);
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

extern "C" int gabo_sphere_pairwise(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2,
                                    int dim, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags,
                                    int diag, gabo_stream_t stream) {
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
        int threads = n2 >= 256 ? 256 : (n2 > 128 ? 192 : (n2 > 64 ? 128 : 64));
        int64_t col_blocks = (n2 + threads - 1) / threads;
        int64_t row_chunks = (n1 + gabo::kSphereRows - 1) / gabo::kSphereRows;
        int64_t nblocks = col_blocks * row_chunks * batch;
        if (nblocks > 0x7fffffffLL) return GABO_ERR_ARG;
        hipLaunchKernelGGL(gabo::sphere_pairwise_kernel, dim3((unsigned)nblocks), dim3(threads), 0, st, x1, x2, out, n1, n2, dim,
                           x1_batch_stride, x2_batch_stride, (int)col_blocks, (int)row_chunks, beta, flags);
    }
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
