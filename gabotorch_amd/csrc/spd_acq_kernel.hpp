// Stand-alone kernel over acq_eval (templates; instantiated per metric in spd_acq.hip / spd_acq_le.hip).
#pragma once
#include "spd_acq_body.hpp"

namespace gabo {

// An error word the HOST can read without a copy: a block that has reported an error (atomicCAS on the device word `status`, inside acq_eval /
// the trust-region bodies) sees it set when it gets here, and repeats it into `mirror` - two ints of mapped host memory, plain stores (no PCIe
// atomics needed; several blocks may write the same values).  Blocks that finished before the failing one write nothing.
static __device__ __forceinline__ void mirror_status(const int* __restrict__ status, int* __restrict__ mirror) {
    if (mirror != nullptr && threadIdx.x == 0) {
        const int e = __atomic_load_n(status, __ATOMIC_RELAXED);
        if (e != 0) {
            mirror[1] = __atomic_load_n(status + 1, __ATOMIC_RELAXED);
            mirror[0] = e;
        }
    }
}

template <int D, int METRIC>
__global__ __launch_bounds__(64) void spd_acq_kernel(const double* __restrict__ x, AcqParams P, double* __restrict__ value,
                                                     double* __restrict__ grad, double* __restrict__ scratch,
                                                     const int* __restrict__ active, int* __restrict__ status, int64_t x_stride,
                                                     int64_t value_stride, double* __restrict__ value_mirror, int* __restrict__ status_mirror) {
    constexpr int T = tri_size(D);
    if (active && active[blockIdx.x] == 0) return;     // masked candidate: outputs left untouched
    __shared__ AcqLds<D> lds;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int64_t i = blockIdx.x;
    // (x_stride / value_stride: the sweep driver keeps its raw samples as rows [value, Mandel vector]; value_mirror: the values once more,
    // densely, in mapped host memory - spd_sweep.hip)
    acq_eval_any<D, METRIC>(x + i * x_stride, P, value + i * value_stride, grad ? grad + i * T : nullptr, scratch ? scratch + i * T * P.n : nullptr,
                            lds, dyn, status, i);
    if (value_mirror != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) value_mirror[i] = value[i * value_stride];
    }
    mirror_status(status, status_mirror);
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
    int64_t x_stride = 0, value_stride = 1;      // 0: the dense layout (d_vec doubles per candidate)
    double* value_mirror = nullptr;
    int* status_mirror = nullptr;                // mapped host copy of an error this launch reports (mirror_status)
};

template <int METRIC, int DMAX>
static int dispatch_acq(const AcqLaunch& a) {
    size_t lds = (size_t)(3 * a.P.n) * sizeof(double);
#define GABO_CASE(DD)                                                                                                                \
    case DD:                                                                                                                         \
        if constexpr (DD <= DMAX) {                                                                                                  \
            hipLaunchKernelGGL((spd_acq_kernel<DD, METRIC>), dim3((unsigned)a.r), dim3(64), lds, a.st, a.x, a.P, a.value, a.grad, a.scratch, \
                               a.active, a.status, a.x_stride ? a.x_stride : (int64_t)(DD * (DD + 1) / 2), a.value_stride, a.value_mirror, a.status_mirror);  \
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
int acq_launch(const AcqLaunch& a);          // by the metric bits of a.P.flags (spd_acq.hip); arguments checked by the caller

}  // namespace gabo
