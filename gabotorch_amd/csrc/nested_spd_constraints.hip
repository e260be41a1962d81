// Eigenvalue bounds of the latent acquisition optimisation of HD-GaBO, stated in the ORIGINAL space (config 5):
//   max_eigenvalue_nested_spd_constraint / min_eigenvalue_nested_spd_constraint   nested_mappings/nested_spd_constraints_utils.py:14-73
// lift the latent point Y (d x d) with projection_from_nested_spd_to_spd (nested_spd_utils.py:51-118),
//   Xrec(Y) = R [[Y, B], [B^T, C]] R^T,  R = [W, V],  B = Y^1/2 K C^1/2,
// and bound lambda_max / lambda_min of the D x D result; the solver differentiates them with respect to Y (symeig(eigenvectors=True) +
// autograd in the reference).  The mapping (W, V, C, K) is fixed during a sweep, so
//   Xrec(Y) = X0 + W Y W^T + (W Y^1/2) P^T + P (W Y^1/2)^T,     X0 = V sym(C) V^T,  P = V (K C^1/2)^T        (gabo_nested_spd_lift_prepare, once)
// and one wave per latent point builds Xrec in LDS, takes BOTH extreme eigenpairs from one Householder reduction (wave_eig_extremes:
// multisection + inverse iteration, 60 k shader cycles at D = 20 where two full eigen-solves cost 2 x 118 k) and pushes v v^T back to Y:
//   d lambda / d Y = a a^T + adj_sqrtm(Y)[a b^T + b a^T],   a = W^T v,  b = P^T v.
// Round 2 evaluated each constraint as ~15 torch launches (sqrtm, block assembly, two products, a full eigen-solve, their adjoints)
// captured into the iteration's hipGraphs.
#include "gabo_device.hpp"
#include "lds_linalg.hpp"
#include "nested_spd_lift.hpp"
#include "../../include/gabo_hip.h"

namespace gabo {

// one block: X0 = V sym(C) V^T, P = V (K C^1/2)^T
template <bool QL>
__global__ __launch_bounds__(256) void nested_spd_lift_prepare_kernel(const double* __restrict__ v, const double* __restrict__ c,
                                                                      const double* __restrict__ k, double* __restrict__ x0,
                                                                      double* __restrict__ p, int D, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = D - d, mm = m * m;
    double* Cl = lds;                 // m x m  sym(C)
    double* Lc = Cl + mm;             // eigenvalues on the diagonal
    double* Uc = Lc + mm;
    double* Cs = Uc + mm;             // C^1/2
    double* Vl = Cs + mm;             // D x m
    double* H = Vl + D * m;           // D x m : V sym(C), later V C^1/2
    double* cs = H + D * m;
    for (int e = threadIdx.x; e < mm; e += blockDim.x) {
        const int r = e / m, cc = e - r * m;
        Cl[e] = 0.5 * (c[r * m + cc] + c[cc * m + r]);
    }
    for (int e = threadIdx.x; e < D * m; e += blockDim.x) Vl[e] = v[e];
    __syncthreads();
    for (int e = threadIdx.x; e < mm; e += blockDim.x) Lc[e] = Cl[e];
    __syncthreads();
    lds_eigh<QL>(Lc, Uc, cs, m);
    lds_fun_from_eig(Lc, Uc, Cs, m, FN_SQRT, cs);
    for (int e = threadIdx.x; e < D * m; e += blockDim.x) {
        const int r = e / m, j = e - r * m;
        double s = 0.0;
        for (int q = 0; q < m; ++q) s = __builtin_fma(Vl[r * m + q], Cl[q * m + j], s);
        H[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
        const int r = e / D, cc = e - r * D;
        if (cc > r) continue;
        double s = 0.0;
        for (int q = 0; q < m; ++q) s = __builtin_fma(H[r * m + q], Vl[cc * m + q], s);
        x0[r * D + cc] = s;
        x0[cc * D + r] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D * m; e += blockDim.x) {       // H <- V C^1/2
        const int r = e / m, j = e - r * m;
        double s = 0.0;
        for (int q = 0; q < m; ++q) s = __builtin_fma(Vl[r * m + q], Cs[q * m + j], s);
        H[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D * d; e += blockDim.x) {       // P = (V C^1/2) K^T
        const int r = e / d, a = e - r * d;
        double s = 0.0;
        for (int q = 0; q < m; ++q) s = __builtin_fma(H[r * m + q], k[a * m + q], s);
        p[e] = s;
    }
}

// lam[i] = (lambda_max, lambda_min) of Xrec(Y_i); grad[i] (2 x d x d, or null) = their gradients with respect to the symmetric Y_i
template <bool QLD, bool QLd>
__global__ __launch_bounds__(64) void nested_spd_extremes_kernel(const double* __restrict__ y, const double* __restrict__ w,
                                                                 const double* __restrict__ p, const double* __restrict__ x0,
                                                                 double* __restrict__ lam, double* __restrict__ grad, int D, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const size_t i = blockIdx.x;
    const int dd = d * d;
    const NestedExtremesOut out = nested_extremes_body<QLD, QLd>(y + i * dd, w, p, x0, D, d, lds, grad != nullptr);
    if (threadIdx.x < 2) lam[i * 2 + threadIdx.x] = out.lam[threadIdx.x];
    if (grad != nullptr)
        for (int e = threadIdx.x; e < 2 * dd; e += 64) grad[i * 2 * dd + e] = out.grad[e];
}

}  // namespace gabo

extern "C" {

int gabo_nested_spd_lift_prepare(const double* v, const double* c, const double* k, double* x0, double* p, int D, int d, gabo_stream_t stream) {
    if (D < 2 || D > GABO_SPD_MAX_DIM || d < 1 || d >= D) return GABO_ERR_DIM;
    if (!v || !c || !k || !x0 || !p) return GABO_ERR_ARG;
    const int m = D - d;
    const size_t lds = (size_t)(4 * m * m + 2 * D * m + gabo::kJacobiScratch) * sizeof(double);
    if (m >= gabo::kWaveEighMinDim)
        hipLaunchKernelGGL(gabo::nested_spd_lift_prepare_kernel<true>, dim3(1), dim3(256), lds, (hipStream_t)stream, v, c, k, x0, p, D, d);
    else
        hipLaunchKernelGGL(gabo::nested_spd_lift_prepare_kernel<false>, dim3(1), dim3(256), lds, (hipStream_t)stream, v, c, k, x0, p, D, d);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_nested_spd_extreme_eigenvalues(const double* y, const double* w, const double* p, const double* x0, double* lam, double* grad,
                                        int64_t R, int D, int d, gabo_stream_t stream) {
    if (D < 2 || D > GABO_SPD_MAX_DIM || d < 1 || d >= D) return GABO_ERR_DIM;
    if (R < 0) return GABO_ERR_ARG;
    if (R == 0) return GABO_OK;
    if (!y || !w || !p || !x0 || !lam || R > 0x7fffffffLL) return GABO_ERR_ARG;
    const size_t lds = gabo::nested_extremes_lds_doubles(D, d) * sizeof(double);
    const bool qD = D >= gabo::kWaveEighMinDim, qd = d >= gabo::kWaveEighMinDim;
#define GABO_NC_LAUNCH(A, B) \
    hipLaunchKernelGGL((gabo::nested_spd_extremes_kernel<A, B>), dim3((unsigned)R), dim3(64), lds, (hipStream_t)stream, y, w, p, x0, lam, grad, D, d)
    if (qD && qd) GABO_NC_LAUNCH(true, true);
    else if (qD) GABO_NC_LAUNCH(true, false);
    else GABO_NC_LAUNCH(false, false);
#undef GABO_NC_LAUNCH
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}
}
