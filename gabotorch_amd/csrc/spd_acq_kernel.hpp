// Stand-alone kernel over acq_eval (templates; instantiated per metric in spd_acq.hip / spd_acq_le.hip).
#pragma once
#include "spd_acq_body.hpp"

namespace gabo {

template <int D, int METRIC>
__global__ __launch_bounds__(64) void spd_acq_kernel(const double* __restrict__ x, AcqParams P, double* __restrict__ value,
                                                     double* __restrict__ grad, double* __restrict__ scratch,
                                                     const int* __restrict__ active, int* __restrict__ status) {
    constexpr int T = tri_size(D);
    if (active && active[blockIdx.x] == 0) return;     // masked candidate: outputs left untouched
    __shared__ AcqLds<D> lds;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    acq_eval_any<D, METRIC>(x + i * T, P, value + i, grad ? grad + i * T : nullptr, scratch ? scratch + i * T * P.n : nullptr, lds, dyn,
                            status, i);
}

struct AcqLaunch {
    const double* x;
    AcqParams P;
    double *value, *grad, *scratch;
    int64_t r;
    int d;
    const int* active;
    int* status;
    hipStream_t st;
};

template <int METRIC, int DMAX>
static int dispatch_acq(const AcqLaunch& a) {
    size_t lds = (size_t)(3 * a.P.n) * sizeof(double);
#define GABO_CASE(DD)                                                                                                                \
    case DD:                                                                                                                         \
        if constexpr (DD <= DMAX) {                                                                                                  \
            hipLaunchKernelGGL((spd_acq_kernel<DD, METRIC>), dim3((unsigned)a.r), dim3(64), lds, a.st, a.x, a.P, a.value, a.grad, a.scratch, \
                               a.active, a.status);                                                                                \
            return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;                                                     \
        }                                                                                                                            \
        return GABO_ERR_DIM;
    switch (a.d) {
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

int acq_affine_invariant(const AcqLaunch& a);
int acq_log_euclidean(const AcqLaunch& a);
int acq_frobenius(const AcqLaunch& a);

}  // namespace gabo
