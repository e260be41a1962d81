// Truncated conjugate gradients of the Riemannian trust-region solver on S^d_++, all R restarts in lock step, one wave per
// restart:  robust_trust_regions.py:417-570 (tCG), constrained_trust_regions.py:530-732 (linearised constraints cut the step at
// distance Delta_cons), approximate_hessian.py:11-62 (finite-difference Hessian).  The reference runs this per restart in
// numpy with a full acquisition gradient between the steps; the torch lock-step version (batched_trust_regions.py) spends its
// time in ~100 tiny element-wise launches per step.  Here one inner iteration is
//     gabo_spd_tcg_fd_point  ->  [acquisition gradient at the FD point: kernel strip, GP+EI, kernel backward]  ->  gabo_spd_tcg_step
// Everything is kept in WHITENED coordinates u~ = L^-1 u L^-T (L = chol(x), x fixed during tCG), where the affine-invariant
// metric tr(x^-1 u x^-1 v) is the Frobenius dot, retr(x, u) = L expm(u~) L^T and the Riemannian gradient at x1 = L E L^T
// transported back to x (identity transport) is  E L^T sym(egrad) L E.
#include "spd_tcg_body.hpp"

namespace gabo {

__global__ __launch_bounds__(64) void spd_tcg_begin_kernel(const double* __restrict__ x, const double* __restrict__ g,
                                                           const double* __restrict__ gc, const double* __restrict__ fc,
                                                           const uint8_t* __restrict__ active, const double* __restrict__ delta_tr,
                                                           void* wsbase, int64_t R, int d, int C, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t i = blockIdx.x;
    TcgWs w = tcg_layout(wsbase, R, d, C);
    if (i == 0 && threadIdx.x == 0) { w.counters[0] = 0; w.counters[1] = 0; w.counters[3] = 0; }
    tcg_begin(x + i * d * d, g + i * d * d, gc, fc, active[i] != 0, delta_tr[i], w, i, R, d, C, status, lds);
}

// counters[3] = 1: the steps that follow run without the preconditioner (use_rand)
__global__ __launch_bounds__(64) void spd_tcg_begin_rand_kernel(void* wsbase, const double* __restrict__ eta0, const double* __restrict__ heta0,
                                                                int64_t R, int d, int C) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t i = blockIdx.x;
    TcgWs w = tcg_layout(wsbase, R, d, C);
    if (i == 0 && threadIdx.x == 0) w.counters[3] = 1;
    tcg_begin_rand(w, i, R, d, C, eta0 + i * d * d, heta0 + i * d * d, lds);
}

__global__ __launch_bounds__(64) void spd_tcg_fd_point_kernel(void* wsbase, double* __restrict__ x_fd, int64_t R, int d, int C) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t i = blockIdx.x;
    TcgWs w = tcg_layout(wsbase, R, d, C);
    if (i == 0 && threadIdx.x == 0) w.counters[0] = 0;       // any_running, set again by the step kernel
    tcg_fd_point(w, i, d, x_fd + i * (int64_t)(d * (d + 1) / 2), lds);
}

__global__ __launch_bounds__(64) void spd_tcg_step_kernel(void* wsbase, const double* __restrict__ egrad_fd, int64_t R, int d, int C,
                                                          int neq, double delta_cons, double theta, double kappa, int mininner) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t i = blockIdx.x;
    TcgWs w = tcg_layout(wsbase, R, d, C);
    const int iter = w.counters[1];
    if (i == 0 && threadIdx.x == 0) w.counters[2] = iter + 1;   // published to counters[1] by spd_tcg_advance_kernel
    const bool running = tcg_step(w, i, R, d, C, egrad_fd + i * (int64_t)(d * (d + 1) / 2), neq, delta_cons, theta, kappa, mininner,
                                  iter, lds, w.counters[3] != 0);
    if (running && threadIdx.x == 0) atomicOr(w.counters, 1);
}

// publishes the inner-iteration index written by the step kernel (one thread; ordered by the stream)
__global__ void spd_tcg_advance_kernel(void* wsbase, int64_t R, int d, int C, int* __restrict__ any_running_out) {
    TcgWs w = tcg_layout(wsbase, R, d, C);
    w.counters[1] = w.counters[2];
    if (any_running_out) *any_running_out = w.counters[0];
}

__global__ __launch_bounds__(64) void spd_tcg_end_kernel(void* wsbase, double* __restrict__ eta, double* __restrict__ heta,
                                                         int* __restrict__ stop, int64_t R, int d, int C) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t i = blockIdx.x;
    TcgWs w = tcg_layout(wsbase, R, d, C);
    tcg_end(w, i, d, eta + i * d * d, heta + i * d * d, lds);
    if (threadIdx.x == 0) stop[i] = w.stop[i];
}

}  // namespace gabo

extern "C" {

size_t gabo_spd_tcg_workspace_bytes(int64_t r, int d, int n_constraints) {
    if (r < 0 || d < 1 || n_constraints < 0) return 0;
    return gabo::tcg_layout(nullptr, r, d, n_constraints).bytes;
}

size_t gabo_spd_tcg_running_offset(int64_t r, int d, int n_constraints) {
    if (r < 0 || d < 1 || n_constraints < 0) return 0;
    return (size_t)((char*)gabo::tcg_layout(nullptr, r, d, n_constraints).running - (char*)nullptr);
}

static int tcg_args_ok(int64_t r, int d, int c) {
    if (d < 1 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (r < 0 || r > 0x7fffffffLL || c < 0 || c > gabo::kMaxCons) return GABO_ERR_ARG;
    return GABO_OK;
}

int gabo_spd_tcg_begin(const double* x, const double* grad, const double* cons_grads, const double* cons_values,
                       const uint8_t* active, const double* trust_radius, void* workspace, size_t workspace_bytes, int64_t r, int d,
                       int n_constraints, int* status, gabo_stream_t stream) {
    int rc = tcg_args_ok(r, d, n_constraints);
    if (rc != GABO_OK) return rc;
    if (r == 0) return GABO_OK;
    if (!x || !grad || !active || !trust_radius || !workspace || (n_constraints > 0 && (!cons_grads || !cons_values)))
        return GABO_ERR_ARG;
    if (workspace_bytes < gabo_spd_tcg_workspace_bytes(r, d, n_constraints)) return GABO_ERR_ARG;
    size_t lds = (size_t)(5 * d * d) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_tcg_begin_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, x, grad, cons_grads,
                       cons_values, active, trust_radius, workspace, r, d, n_constraints, status);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_tcg_begin_rand(void* workspace, const double* eta0, const double* heta0, int64_t r, int d, int n_constraints,
                            gabo_stream_t stream) {
    int rc = tcg_args_ok(r, d, n_constraints);
    if (rc != GABO_OK) return rc;
    if (r == 0) return GABO_OK;
    if (!workspace || !eta0 || !heta0) return GABO_ERR_ARG;
    size_t lds = (size_t)(5 * d * d) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_tcg_begin_rand_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, workspace, eta0, heta0, r, d,
                       n_constraints);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_tcg_fd_point(void* workspace, double* x_fd_mandel, int64_t r, int d, int n_constraints, gabo_stream_t stream) {
    int rc = tcg_args_ok(r, d, n_constraints);
    if (rc != GABO_OK) return rc;
    if (r == 0) return GABO_OK;
    if (!workspace || !x_fd_mandel) return GABO_ERR_ARG;
    size_t lds = (size_t)(4 * d * d + 2) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_tcg_fd_point_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, workspace, x_fd_mandel,
                       r, d, n_constraints);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_tcg_step(void* workspace, const double* egrad_fd_mandel, int* any_running, int64_t r, int d, int n_constraints,
                      int n_equalities, double delta_cons, double theta, double kappa, int mininner, gabo_stream_t stream) {
    int rc = tcg_args_ok(r, d, n_constraints);
    if (rc != GABO_OK) return rc;
    if (r == 0) return GABO_OK;
    if (!workspace || !egrad_fd_mandel || n_equalities < 0 || n_equalities > n_constraints) return GABO_ERR_ARG;
    size_t lds = (size_t)(5 * d * d) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_tcg_step_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, workspace, egrad_fd_mandel,
                       r, d, n_constraints, n_equalities, delta_cons, theta, kappa, mininner);
    hipLaunchKernelGGL(gabo::spd_tcg_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, workspace, r, d, n_constraints,
                       any_running);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int gabo_spd_tcg_end(void* workspace, double* eta, double* heta, int* stop_reason, int64_t r, int d, int n_constraints,
                     gabo_stream_t stream) {
    int rc = tcg_args_ok(r, d, n_constraints);
    if (rc != GABO_OK) return rc;
    if (r == 0) return GABO_OK;
    if (!workspace || !eta || !heta || !stop_reason) return GABO_ERR_ARG;
    size_t lds = (size_t)(4 * d * d) * sizeof(double);
    hipLaunchKernelGGL(gabo::spd_tcg_end_kernel, dim3((unsigned)r), dim3(64), lds, (hipStream_t)stream, workspace, eta, heta,
                       stop_reason, r, d, n_constraints);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

}  // extern "C"
