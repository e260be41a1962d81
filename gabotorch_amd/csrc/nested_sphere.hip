// Nested-sphere projection S^d -> S^(d-1) (principal nested spheres), the per-point part:
//   projection_from_sphere_to_nested_sphere / _to_next_subsphere     nested_mappings/nested_spheres_utils.py:13-114
//   used by NestedSphereGaussianKernel.forward                        kernel_utils/kernels_nested_sphere.py:125-152
// The rotation of the axis to the north pole is one d x d matrix shared by all points, so U = X R^T is a plain GEMM (done by the
// caller with rocBLAS through torch.matmul); what is specific is the per-point epilogue, one lane per point:
//   theta = acos(clamp(U[d-1]))                                        (distance to the rotated axis, sphere_utils_torch.py:52-55)
//   mode 0 (next subsphere):  w = sin r / ((sin theta + 1e-6)(sin r + 1e-6)) U[0:d-1],   z = w / (|w| + 1e-6)
//   mode 1 (nested sphere, still rotated): y = (sin r U + sin(theta - r) e_d) / (sin theta + 1e-6)
// and the backward of mode 0 (the reference differentiates it by autograd for the GP fit / acquisition gradient).
// The reference rotates back to the axis and forward again between the two steps of mode 0; R[:-1] R^T = [I 0] makes that the
// identity on the first d-1 coordinates, which is what is used here.
#include "gabo_device.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

constexpr double kNsEps = 1e-6;                 // the reference's "+ 1e-6 * ones" (nested_spheres_utils.py:56,105,110)
constexpr double kNsClamp = 1.0 - 1e-15;        // sphere_utils_torch.py:53

__global__ __launch_bounds__(256) void nested_sphere_epilogue_kernel(const double* __restrict__ u, double* __restrict__ out, int64_t n,
                                                                     int d, double r, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* U = u + i * d;
    double t = U[d - 1];
    t = t > kNsClamp ? kNsClamp : (t < -kNsClamp ? -kNsClamp : t);
    const double theta = acos(t);
    const double st = sin(theta);
    const double sr = sin(r);
    if (mode == 1) {
        double* O = out + i * d;
        const double inv = 1.0 / (st + kNsEps);
        for (int k = 0; k < d - 1; ++k) O[k] = sr * U[k] * inv;
        O[d - 1] = (sr * U[d - 1] + sin(theta - r)) * inv;
        return;
    }
    double* O = out + i * (d - 1);
    const double a = sr / ((st + kNsEps) * (sr + kNsEps));
    double nn = 0.0;
    for (int k = 0; k < d - 1; ++k) { double w = a * U[k]; nn = __builtin_fma(w, w, nn); }
    const double inv = 1.0 / (__builtin_sqrt(nn) + kNsEps);
    for (int k = 0; k < d - 1; ++k) O[k] = a * U[k] * inv;
}

__global__ __launch_bounds__(256) void nested_sphere_epilogue_backward_kernel(const double* __restrict__ u, const double* __restrict__ gz,
                                                                              double* __restrict__ gu, int64_t n, int d, double r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* U = u + i * d;
    const double* G = gz + i * (d - 1);
    double* O = gu + i * d;
    const double tl = U[d - 1];
    const bool inside = tl < kNsClamp && tl > -kNsClamp;       // clamp passes the gradient only strictly inside
    const double t = tl > kNsClamp ? kNsClamp : (tl < -kNsClamp ? -kNsClamp : tl);
    const double theta = acos(t);
    const double st = sin(theta);
    const double sr = sin(r);
    const double a = sr / ((st + kNsEps) * (sr + kNsEps));
    double nn = 0.0, gw_dot = 0.0;
    for (int k = 0; k < d - 1; ++k) {
        double w = a * U[k];
        nn = __builtin_fma(w, w, nn);
        gw_dot = __builtin_fma(G[k], w, gw_dot);
    }
    const double nrm = __builtin_sqrt(nn);
    const double den = nrm + kNsEps;
    // z = w / den:  g_w = g_z / den - (g_z . w) w / (|w| den^2)
    const double coef = nrm > 0.0 ? gw_dot / (nrm * den * den) : 0.0;
    double ga = 0.0;
    for (int k = 0; k < d - 1; ++k) {
        double w = a * U[k];
        double gw = G[k] / den - coef * w;
        O[k] = a * gw;
        ga = __builtin_fma(gw, U[k], ga);
    }
    // a(theta) = sr / ((sin theta + eps)(sr + eps)):  da/dtheta = -a cos theta / (sin theta + eps);  dtheta/dt = -1 / sqrt(1 - t^2)
    O[d - 1] = inside ? ga * a * cos(theta) / ((st + kNsEps) * __builtin_sqrt((1.0 - t) * (1.0 + t))) : 0.0;
}

}  // namespace gabo

extern "C" {

int gabo_nested_sphere_epilogue(const double* rotated, double* out, int64_t n, int d, double dist_to_axis, int mode,
                                gabo_stream_t stream) {
    if (d < 2 || d > 4096) return GABO_ERR_DIM;
    if (n < 0 || (mode != 0 && mode != 1)) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    if (!rotated || !out) return GABO_ERR_ARG;
    hipLaunchKernelGGL(gabo::nested_sphere_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rotated, out, n, d, dist_to_axis, mode);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_nested_sphere_epilogue_backward(const double* rotated, const double* grad_out, double* grad_rotated, int64_t n, int d,
                                         double dist_to_axis, gabo_stream_t stream) {
    if (d < 2 || d > 4096) return GABO_ERR_DIM;
    if (n < 0) return GABO_ERR_ARG;
    if (n == 0) return GABO_OK;
    if (!rotated || !grad_out || !grad_rotated) return GABO_ERR_ARG;
    hipLaunchKernelGGL(gabo::nested_sphere_epilogue_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, rotated, grad_out, grad_rotated, n, d, dist_to_axis);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // extern "C"
