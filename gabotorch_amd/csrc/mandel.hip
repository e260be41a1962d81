// Mandel vector <-> symmetric matrix (Riemannian_utils/spd_utils_torch.py:159-226) as flat, fully coalesced element-wise
// kernels: one lane per OUTPUT element, closed-form index map instead of the reference's per-matrix Python loop.
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

__global__ __launch_bounds__(256) void mandel_to_matrix_kernel(const double* __restrict__ vec, double* __restrict__ mat, int64_t n,
                                                               int d) {
    const int64_t dd = (int64_t)d * d;
    const int dv = d * (d + 1) / 2;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n * dd; g += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = g / dd;
        int rc = (int)(g - q * dd);
        int r = rc / d, c = rc - r * d;
        int hi = r > c ? r : c, lo = r > c ? c : r;
        double v = vec[q * dv + mandel_pos(d, hi, lo)];
        mat[g] = (r == c) ? v : v / kSqrt2;  // spd_utils_torch.py:186-187
    }
}

__global__ __launch_bounds__(256) void matrix_to_mandel_kernel(const double* __restrict__ mat, double* __restrict__ vec, int64_t n,
                                                               int d) {
    const int64_t dd = (int64_t)d * d;
    const int dv = d * (d + 1) / 2;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n * dv; g += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = g / dv;
        int e = (int)(g - q * dv);
        // which diagonal k holds entry e: entries before diagonal k = k*d - k(k-1)/2
        int k = 0;
        while (k + 1 < d && (k + 1) * d - (k + 1) * k / 2 <= e) ++k;
        int c = e - (k * d - k * (k - 1) / 2);
        int r = c + k;
        const double* m = mat + q * dd;
        if (k == 0) {
            vec[g] = m[r * d + c];
        } else {
            // 0.5 * (sqrt2*upper + sqrt2*lower): both triangles, so an autograd gradient comes out symmetric (:219)
            vec[g] = 0.5 * (kSqrt2 * m[c * d + r] + kSqrt2 * m[r * d + c]);
        }
    }
}

}  // namespace gabo

extern "C" {

int gabo_mandel_to_matrix(const double* vec, double* mat, int64_t n, int d, gabo_stream_t stream) {
    if (n < 0) return GABO_ERR_ARG;
    if (d < 1 || d > 64) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    if (!vec || !mat) return GABO_ERR_ARG;
    int64_t tot = n * d * d;
    int64_t blocks = (tot + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gabo::mandel_to_matrix_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vec, mat, n, d);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_matrix_to_mandel(const double* mat, double* vec, int64_t n, int d, gabo_stream_t stream) {
    if (n < 0) return GABO_ERR_ARG;
    if (d < 1 || d > 64) return GABO_ERR_DIM;
    if (n == 0) return GABO_OK;
    if (!vec || !mat) return GABO_ERR_ARG;
    int64_t tot = n * (d * (d + 1) / 2);
    int64_t blocks = (tot + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gabo::matrix_to_mandel_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mat, vec, n, d);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_version(void) { return 100; }
}
