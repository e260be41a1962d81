// Sphere pairwise kernel matrix on gfx950:  out_ij = f(acos(clamp(<x1_i, x2_j>, -1+1e-15, 1-1e-15)))
// replaces kernel_utils/kernels_sphere.py:71-94,118-134 + Riemannian_utils/sphere_utils_torch.py:12-55.
//
// The reference materialises two N1 x N2 x dim tensors and a degenerate bmm; here a block owns a ROWS x 256 output
// tile.  Lane = column j (so the fp64 stores of a wave are one contiguous 512 B run), each lane keeps ROWS running
// inner products in registers.  The x1 rows of the tile are wave-uniform and come through the scalar cache; the x2
// row of the lane is read once per tile.  HBM traffic = the output matrix (8 B per pair); this kernel is bound by the
// fp64 acos+exp epilogue and the output write, not by operand reads.
#include "gabo_device.hpp"
#include "gabo_mirror.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

constexpr int kSphereRows = 16;   // rows of the output tile per block
constexpr int kSphereKC = 16;     // inner-product depth staged through LDS per pass
constexpr int kSphereLd = 257;    // padded leading dimension of the transposed x2 tile: conflict-free transposed writes

template <int MODE>
__device__ __forceinline__ double sphere_finish(double ip, double beta, const MathRegs& mt) {
    const double lo = -1.0 + 1e-15, hi = 1.0 - 1e-15;  // sphere_utils_torch.py:53
    double c = __builtin_fmin(__builtin_fmax(ip, lo), hi);
    double dist = acos_fast(c, mt);                       // :55
    if constexpr (MODE == GABO_OUT_DISTANCE) return dist;
    if constexpr (MODE == GABO_OUT_LAPLACE) return exp_neg(-(dist * beta), mt);
    return exp_neg(-((dist * dist) * beta), mt);          // kernels_sphere.py:91-93
}

// 1-D grid: block id -> (batch, row chunk, column group), column group fastest.  Both operand tiles go through LDS:
// the x2 tile is read from HBM/L2 fully coalesced (its 256 points are contiguous) and stored TRANSPOSED so that lane j
// then reads xs2[k][j] conflict-free; the x1 tile is read back as LDS broadcasts (same address for the whole wave).
// MODE is a template parameter so the 16 row epilogues of a lane are straight-line code the scheduler can interleave:
// each is a long dependent chain (two Horner polynomials), and with ~110 VGPRs only 4 waves per SIMD hide its latency.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void sphere_pairwise_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                              double* __restrict__ out, int64_t n1, int64_t n2, int dim,
                                                              int64_t s1, int64_t s2, int col_blocks, int row_chunks,
                                                              int64_t sym_tiles, double beta, int flags, uint32_t dim_magic) {
    __shared__ double xs2[kSphereKC * kSphereLd];
    __shared__ double xs1[kSphereRows * kSphereKC];
    const int tid = threadIdx.x;
    int64_t cg, rc, b;
    if (flags & GABO_SYMMETRIC) {       // x1 is x2: only tiles touching the upper triangle exist (see spd_pairwise.hip)
        b = blockIdx.x / sym_tiles;
        int64_t t = blockIdx.x - b * sym_tiles;
        cg = 0;
        for (;;) {
            int64_t cnt = sym_chunks_of(cg, blockDim.x, kSphereRows, row_chunks);
            if (t < cnt) break;
            t -= cnt;
            ++cg;
        }
        rc = t;
    } else {
        const int64_t bid = blockIdx.x;
        cg = bid % col_blocks;
        rc = (bid / col_blocks) % row_chunks;
        b = bid / ((int64_t)col_blocks * row_chunks);
    }
    const int64_t j0 = cg * blockDim.x;
    const int64_t j = j0 + tid;
    const int64_t i0 = rc * kSphereRows;
    const int ncols = (int)((n2 - j0 < (int64_t)blockDim.x) ? n2 - j0 : (int64_t)blockDim.x);
    const int nrows = (int)((n1 - i0 < kSphereRows) ? n1 - i0 : kSphereRows);
    const double* a = x1 + b * s1 + i0 * dim;      // nrows x dim, contiguous
    const double* bt = x2 + b * s2 + j0 * dim;     // ncols x dim, contiguous
    double acc[kSphereRows];
    static_for<kSphereRows>([&](auto r) { acc[decltype(r)::value] = 0.0; });
    for (int k0 = 0; k0 < dim; k0 += kSphereKC) {
        const int kc = dim - k0 < kSphereKC ? dim - k0 : kSphereKC;
        if (k0) __syncthreads();
        if (kc == dim) {
            // whole points fit one pass: the tile is one contiguous run of ncols*dim doubles
            // e / dim by multiply-high with dim_magic = ceil(2^32 / dim) (exact for e < 2^16): a runtime integer division is ~40
            // VALU instructions and there are ten of them per thread here - a fifth of the kernel's instruction count
            for (int e = tid; e < ncols * dim; e += blockDim.x) {
                int jj = dim == 1 ? e : (int)__umulhi((uint32_t)e, dim_magic), kk = e - jj * dim;
                xs2[kk * kSphereLd + jj] = bt[e];
            }
        } else {
            for (int e = tid; e < ncols * kc; e += blockDim.x) {
                int jj = e / kc, kk = e - jj * kc;
                xs2[kk * kSphereLd + jj] = bt[(int64_t)jj * dim + k0 + kk];
            }
        }
        for (int e = tid; e < nrows * kc; e += blockDim.x) {
            int rr = (kc == dim && dim > 1) ? (int)__umulhi((uint32_t)e, dim_magic) : e / kc, kk = e - rr * kc;
            xs1[rr * kSphereKC + kk] = a[(int64_t)rr * dim + k0 + kk];
        }
        __syncthreads();
        for (int kk = 0; kk < kc; ++kk) {
            double y = xs2[kk * kSphereLd + tid];
            static_for<kSphereRows>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                acc[r] = __builtin_fma(xs1[r * kSphereKC + kk], y, acc[r]);
            });
        }
    }
    if (tid < ncols) {
        const MathRegs mt = MathRegs::load();
        double* o = out + b * n1 * n2 + i0 * n2 + j;
        static_for<kSphereRows>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            double val = sphere_finish<MODE>(acc[r], beta, mt);
            if (r < nrows && (!(flags & GABO_SYMMETRIC) || i0 + r <= j)) o[(int64_t)r * n2] = val;
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
    const MathRegs mt = MathRegs::load();
    const int mode = flags & GABO_OUT_MASK;
    out[g] = mode == GABO_OUT_DISTANCE ? sphere_finish<GABO_OUT_DISTANCE>(acc, beta, mt)
                                       : (mode == GABO_OUT_LAPLACE ? sphere_finish<GABO_OUT_LAPLACE>(acc, beta, mt)
                                                                   : sphere_finish<GABO_OUT_GAUSSIAN>(acc, beta, mt));
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
        int64_t sym_tiles = 0;
        if (flags & GABO_SYMMETRIC) {
            if (n1 != n2 || batch > 65535) return GABO_ERR_ARG;
            for (int64_t cg = 0; cg < col_blocks; ++cg) sym_tiles += gabo::sym_chunks_of(cg, threads, gabo::kSphereRows, row_chunks);
        }
        int64_t nblocks = ((flags & GABO_SYMMETRIC) ? sym_tiles : col_blocks * row_chunks) * batch;
        if (nblocks > 0x7fffffffLL) return GABO_ERR_ARG;
        const int mode = flags & GABO_OUT_MASK;
        const uint32_t dim_magic = (uint32_t)((0x100000000ULL + (uint64_t)dim - 1) / (uint64_t)dim);     // ceil(2^32 / dim); dim = 1: unused
#define GABO_SPH_LAUNCH(M)                                                                                                   \
    hipLaunchKernelGGL((gabo::sphere_pairwise_kernel<M>), dim3((unsigned)nblocks), dim3(threads), 0, st, x1, x2, out, n1, n2, \
                       dim, x1_batch_stride, x2_batch_stride, (int)col_blocks, (int)row_chunks, sym_tiles, beta, flags, dim_magic)
        if (mode == GABO_OUT_DISTANCE) GABO_SPH_LAUNCH(GABO_OUT_DISTANCE);
        else if (mode == GABO_OUT_LAPLACE) GABO_SPH_LAUNCH(GABO_OUT_LAPLACE);
        else GABO_SPH_LAUNCH(GABO_OUT_GAUSSIAN);
#undef GABO_SPH_LAUNCH
        if (flags & GABO_SYMMETRIC) {
            int tiles = (int)((n1 + 31) / 32);
            hipLaunchKernelGGL((gabo::mirror_upper_kernel<1>), dim3((unsigned)((int64_t)tiles * (tiles + 1) / 2), (unsigned)batch),
                               dim3(256), 0, st, out, n1, tiles);
        }
    }
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
