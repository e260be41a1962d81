// Host driver of ONE multi-start acquisition sweep on S^d_++ (config 4 of BASELINE.json: gabo_spd, 512 restarts) as two host calls:
//   joint_optimize_manifold -> gen_batch_initial_conditions_manifold (raw samples drawn and scored, manifold_optimize.py:232-321)
//                           -> gen_candidates_manifold (the restarts' local solves, :124-228) -> get_best_candidates (:118-120).
// Every piece of device work it enqueues already exists behind the C ABI - gabo_spd_sample_range, gabo_matrix_to_mandel / gabo_mandel_to_matrix,
// gabo_spd_acq_eval, gabo_spd_manifold_op (egrad2rgrad, norm), gabo_spd_tr_solve - and the Python path (manifold_optimize.py /
// batched_trust_regions.py of this package) issues exactly the same launches in the same order; what this file removes is the ~1.3 ms of
// interpreter time around them (tensor wrappers, allocator, ctypes marshalling: tools/sweep_cprofile.py, tools/solve_host_timeline.py) next to a
// 2.7-ms solve kernel.  The selection heuristic between the two calls (botorch's initialize_q_batch / initialize_q_batch_nonneg [3P], driven by
// torch's generator) stays where it is, on the host in Python, so that the random stream - and with it every selected restart - is the one the
// Python path uses: the two paths return bit-identical candidates (tests/test_gpu_native_sweep.py).
//   gabo_spd_sweep_score: raw samples -> workspace, their acquisition values -> host
//   gabo_spd_sweep_solve: picked samples -> initial value / gradient -> the single-launch trust-region solve -> argmax
// The workspace is the caller's (gabo_spd_sweep_workspace_bytes); nothing is allocated here.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/gabo_hip.h"

namespace gabo {

struct SweepWs {
    double *raw_mat, *raw_mandel, *raw_val;          // max_raw x d x d, max_raw x dv, max_raw
    double *x0_mandel, *x, *xm, *fx, *egm, *eg, *g, *ng, *delta, *cand;
    int64_t *picked, *iters;
    uint8_t* active;
    double* scratch;                                  // r x dv x n (logm spill of gabo_spd_acq_eval with a gradient)
    void* tr;                                         // gabo_spd_tr_workspace_bytes
    size_t tr_bytes, bytes;
};

static SweepWs sweep_layout(void* base, int64_t n, int d, int64_t max_raw, int64_t r, int c) {
    SweepWs w;
    const int64_t dv = (int64_t)d * (d + 1) / 2, dd = (int64_t)d * d;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += (bytes + 255) & ~(size_t)255;
        return q;
    };
    w.raw_mat = (double*)take(max_raw * dd * 8);
    w.raw_mandel = (double*)take(max_raw * dv * 8);
    w.raw_val = (double*)take(max_raw * 8);
    w.x0_mandel = (double*)take(r * dv * 8);
    w.x = (double*)take(r * dd * 8);
    w.xm = (double*)take(r * dv * 8);
    w.fx = (double*)take(r * 8);
    w.egm = (double*)take(r * dv * 8);
    w.eg = (double*)take(r * dd * 8);
    w.g = (double*)take(r * dd * 8);
    w.ng = (double*)take(r * 8);
    w.delta = (double*)take(r * 8);
    w.cand = (double*)take(r * dv * 8);
    w.picked = (int64_t*)take(r * 8);
    w.iters = (int64_t*)take(r * 8);
    w.active = (uint8_t*)take(r);
    w.scratch = (double*)take((size_t)r * dv * n * 8);
    w.tr_bytes = gabo_spd_tr_workspace_bytes(r, d, c, n);
    w.tr = take(w.tr_bytes + 8);
    w.bytes = (size_t)(p - (char*)base);
    return w;
}

__global__ __launch_bounds__(256) void sweep_gather_kernel(const double* __restrict__ src, const int64_t* __restrict__ idx, double* __restrict__ dst,
                                                           int64_t r, int width) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= r * width) return;
    const int64_t row = e / width;
    dst[e] = src[idx[row] * width + (e - row * width)];
}

__global__ __launch_bounds__(256) void sweep_init_kernel(double* __restrict__ delta, uint8_t* __restrict__ active, int64_t* __restrict__ iters,
                                                         int64_t r, double delta0) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= r) return;
    delta[e] = delta0;
    active[e] = 1;
    iters[e] = 0;
}

}  // namespace gabo

extern "C" size_t gabo_spd_sweep_workspace_bytes(int64_t n_train, int d, int64_t max_raw, int64_t restarts, int n_constraints) {
    if (n_train < 1 || d < 2 || max_raw < 0 || restarts < 0 || n_constraints < 0) return 0;
    return gabo::sweep_layout(nullptr, n_train, d, max_raw, restarts, n_constraints).bytes;
}

extern "C" int gabo_spd_sweep_score(const gabo_spd_sweep_config* cfg, int64_t count, int64_t max_raw, int64_t restarts, uint64_t seed,
                                    const double* raw_matrices_host, double* values_host, void* workspace, size_t workspace_bytes, int* status,
                                    gabo_stream_t stream) {
    if (!cfg || !values_host || !workspace || !status || count < 1 || count > max_raw || restarts < 1) return GABO_ERR_ARG;
    const int d = cfg->d;
    if (d < 2 || d > 8 || cfg->n_constraints < 0 || cfg->n_constraints > GABO_SWEEP_MAX_CONSTRAINTS) return GABO_ERR_DIM;
    const gabo_spd_acq_params& a = cfg->acq;
    const gabo::SweepWs w = gabo::sweep_layout(workspace, a.n, d, max_raw, restarts, cfg->n_constraints);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // spd_sample on the device (manifolds.PositiveDefinite.rand_batch_device), the Mandel map of the post-processing (manifold_optimize.py:291),
    // the acquisition value of every raw sample (the fused chain's cost, sign -1: `values` below are the acquisition values themselves)
    if (raw_matrices_host) {
        // the caller's own sampler (`manifold.rand` is user code: the reference binds spd_sample to it, examples/gabo_spd.py:102) drew them on the host
        if (hipMemcpyAsync(w.raw_mat, raw_matrices_host, (size_t)count * d * d * 8, hipMemcpyHostToDevice, st) != hipSuccess) return GABO_ERR_LAUNCH;
    } else if ((rc = gabo_spd_sample_range(w.raw_mat, 0, count, d, cfg->min_eig, cfg->max_eig, seed, 0, stream)) != GABO_OK) {
        return rc;
    }
    if ((rc = gabo_matrix_to_mandel(w.raw_mat, w.raw_mandel, count, d, stream)) != GABO_OK) return rc;
    if ((rc = gabo_spd_acq_eval(w.raw_mandel, a.train_factors, a.alpha, a.linv, a.linv_t, w.raw_val, nullptr, nullptr, count, a.n, d, a.beta, a.flags,
                                a.mean, a.outputscale, a.kxx, a.best_f, a.kind, a.maximize, 1.0, nullptr, status, stream)) != GABO_OK)
        return rc;
    if (hipMemcpyAsync(values_host, w.raw_val, (size_t)count * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    return GABO_OK;
}

extern "C" int gabo_spd_sweep_solve(const gabo_spd_sweep_config* cfg, const int64_t* picked_host, int64_t restarts, int64_t max_raw,
                                    int64_t* best_index_host, double* best_value_host, int64_t* max_iterations_host, double** candidates_dev,
                                    double** cost_dev, int64_t** iterations_dev, void* workspace, size_t workspace_bytes, int* status,
                                    gabo_stream_t stream) {
    if (!cfg || !picked_host || !best_index_host || !best_value_host || !workspace || !status || restarts < 1) return GABO_ERR_ARG;
    const int d = cfg->d, c = cfg->n_constraints;
    if (d < 2 || d > 8 || c < 0 || c > GABO_SWEEP_MAX_CONSTRAINTS) return GABO_ERR_DIM;
    const gabo_spd_acq_params& a = cfg->acq;
    const int64_t r = restarts, dv = (int64_t)d * (d + 1) / 2;
    for (int64_t k = 0; k < r; ++k)
        if (picked_host[k] < 0 || picked_host[k] >= max_raw) return GABO_ERR_ARG;
    const gabo::SweepWs w = gabo::sweep_layout(workspace, a.n, d, max_raw, r, c);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (hipMemcpyAsync(w.picked, picked_host, (size_t)r * 8, hipMemcpyHostToDevice, st) != hipSuccess) return GABO_ERR_LAUNCH;
    // the chosen raw samples (Mandel) -> matrices (pre_processing_manifold, manifold_optimize.py:170-171) -> Mandel again for the evaluation
    // (post_processing_manifold inside the cost, :177-180): the same two maps the Python path applies, so the operands carry the same bits
    hipLaunchKernelGGL(gabo::sweep_gather_kernel, dim3((unsigned)((r * dv + 255) / 256)), dim3(256), 0, st, w.raw_mandel, w.picked, w.x0_mandel, r, (int)dv);
    if ((rc = gabo_mandel_to_matrix(w.x0_mandel, w.x, r, d, stream)) != GABO_OK) return rc;
    if ((rc = gabo_matrix_to_mandel(w.x, w.xm, r, d, stream)) != GABO_OK) return rc;
    // cost (= -acquisition) and its Euclidean gradient at the starts, then the Riemannian gradient and its norm ([3P] egrad2rgrad / norm of
    // pymanopt's PositiveDefinite, robust_trust_regions.py:148-158)
    if ((rc = gabo_spd_acq_eval(w.xm, a.train_factors, a.alpha, a.linv, a.linv_t, w.fx, w.egm, w.scratch, r, a.n, d, a.beta, a.flags, a.mean,
                                a.outputscale, a.kxx, a.best_f, a.kind, a.maximize, -1.0, nullptr, status, stream)) != GABO_OK)
        return rc;
    if ((rc = gabo_mandel_to_matrix(w.egm, w.eg, r, d, stream)) != GABO_OK) return rc;
    if ((rc = gabo_spd_manifold_op(GABO_SPD_EGRAD2RGRAD, w.x, w.eg, nullptr, nullptr, w.g, nullptr, r, d, status, stream)) != GABO_OK) return rc;
    if ((rc = gabo_spd_manifold_op(GABO_SPD_NORM, w.x, w.g, nullptr, nullptr, w.ng, nullptr, r, d, status, stream)) != GABO_OK) return rc;
    hipLaunchKernelGGL(gabo::sweep_init_kernel, dim3((unsigned)((r + 255) / 256)), dim3(256), 0, st, w.delta, w.active, w.iters, r, cfg->delta0);
    if (hipMemsetAsync(w.tr, 0, w.tr_bytes, st) != hipSuccess) return GABO_ERR_LAUNCH;
    gabo_spd_acq_params acq = a;
    acq.out_sign = -1.0;
    if ((rc = gabo_spd_tr_solve(w.x, w.fx, w.g, w.ng, w.delta, w.active, w.iters, &acq, c, cfg->constraint_kind, cfg->constraint_bound, cfg->strict,
                                w.tr, w.tr_bytes, r, d, cfg->delta_cons, cfg->theta, cfg->kappa, cfg->mininner, cfg->maxinner, cfg->delta_bar,
                                cfg->rho_prime, cfg->rho_regularization, cfg->mingradnorm, cfg->maxiter, nullptr, nullptr, nullptr, 0, status,
                                stream)) != GABO_OK)
        return rc;
    if ((rc = gabo_matrix_to_mandel(w.x, w.cand, r, d, stream)) != GABO_OK) return rc;
    // get_best_candidates (manifold_optimize.py:118-120): argmax of the acquisition value = argmin of the cost, first index on ties
    std::vector<double> fx((size_t)r);
    std::vector<int64_t> it((size_t)r);
    if (hipMemcpyAsync(fx.data(), w.fx, (size_t)r * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipMemcpyAsync(it.data(), w.iters, (size_t)r * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    int64_t best = 0, maxit = 0;
    for (int64_t k = 0; k < r; ++k) {
        // (torch.argmax semantics: a NaN value wins; the first of equal values)
        const double v = -fx[(size_t)k], b = -fx[(size_t)best];
        if ((v > b && b == b) || (v != v && b == b)) best = k;
        if (it[(size_t)k] > maxit) maxit = it[(size_t)k];
    }
    *best_index_host = best;
    *best_value_host = -fx[(size_t)best];
    if (max_iterations_host) *max_iterations_host = maxit;
    if (candidates_dev) *candidates_dev = w.cand;
    if (cost_dev) *cost_dev = w.fx;
    if (iterations_dev) *iterations_dev = w.iters;
    return GABO_OK;
}

// ---- the sphere twin: gen_batch_initial_conditions_manifold + gen_candidates_manifold + get_best_candidates on S^(dim-1) ----------------------------
// The reference's gabo_sphere examples (examples/bo_sphere/benchmark_examples/gabo_sphere.py:151-175): stock TrustRegions, no constraints, exact or
// finite-difference Hessian-vector products.  Same two-call shape as above; the raw samples always come from the caller's host sampler (`manifold.rand`).
namespace gabo {

struct SphSweepWs {
    double *raw, *raw_val, *x, *fx, *eg, *g, *ng, *delta;
    int64_t *picked, *iters;
    uint8_t* active;
    void* tr;
    size_t tr_bytes, bytes;
};

static SphSweepWs sph_sweep_layout(void* base, int dim, int64_t max_raw, int64_t r) {
    SphSweepWs w;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += (bytes + 255) & ~(size_t)255;
        return q;
    };
    w.raw = (double*)take((size_t)max_raw * dim * 8);
    w.raw_val = (double*)take((size_t)max_raw * 8);
    w.x = (double*)take((size_t)r * dim * 8);
    w.fx = (double*)take((size_t)r * 8);
    w.eg = (double*)take((size_t)r * dim * 8);
    w.g = (double*)take((size_t)r * dim * 8);
    w.ng = (double*)take((size_t)r * 8);
    w.delta = (double*)take((size_t)r * 8);
    w.picked = (int64_t*)take((size_t)r * 8);
    w.iters = (int64_t*)take((size_t)r * 8);
    w.active = (uint8_t*)take((size_t)r);
    w.tr_bytes = gabo_sphere_tr_workspace_bytes(r, dim, 0);
    w.tr = take(w.tr_bytes + 8);
    w.bytes = (size_t)(p - (char*)base);
    return w;
}

// ||g_i|| of r tangent vectors ([3P] pymanopt Sphere.norm = the Euclidean norm)
__global__ __launch_bounds__(256) void sweep_rownorm_kernel(const double* __restrict__ g, double* __restrict__ out, int64_t r, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r) return;
    double s = 0.0;
    for (int k = 0; k < dim; ++k) s += g[i * dim + k] * g[i * dim + k];
    out[i] = __builtin_sqrt(s);
}

}  // namespace gabo

extern "C" size_t gabo_sphere_sweep_workspace_bytes(int dim, int64_t max_raw, int64_t restarts) {
    if (dim < 2 || max_raw < 0 || restarts < 0) return 0;
    return gabo::sph_sweep_layout(nullptr, dim, max_raw, restarts).bytes;
}

extern "C" int gabo_sphere_sweep_score(const gabo_sphere_sweep_config* cfg, int64_t count, int64_t max_raw, int64_t restarts,
                                       const double* raw_points_host, double* values_host, void* workspace, size_t workspace_bytes,
                                       gabo_stream_t stream) {
    if (!cfg || !raw_points_host || !values_host || !workspace || count < 1 || count > max_raw || restarts < 1) return GABO_ERR_ARG;
    const int dim = cfg->acq.dim;
    if (dim < 2) return GABO_ERR_DIM;
    const gabo::SphSweepWs w = gabo::sph_sweep_layout(workspace, dim, max_raw, restarts);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(w.raw, raw_points_host, (size_t)count * dim * 8, hipMemcpyHostToDevice, st) != hipSuccess) return GABO_ERR_LAUNCH;
    gabo_sphere_acq_params acq = cfg->acq;
    acq.out_sign = 1.0;
    int rc;
    if ((rc = gabo_sphere_acq_eval(w.raw, &acq, w.raw_val, nullptr, count, stream)) != GABO_OK) return rc;
    if (hipMemcpyAsync(values_host, w.raw_val, (size_t)count * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    return GABO_OK;
}

extern "C" int gabo_sphere_sweep_solve(const gabo_sphere_sweep_config* cfg, const int64_t* picked_host, int64_t restarts, int64_t max_raw,
                                       int64_t* best_index_host, double* best_value_host, int64_t* max_iterations_host, double** candidates_dev,
                                       double** cost_dev, int64_t** iterations_dev, void* workspace, size_t workspace_bytes,
                                       gabo_stream_t stream) {
    if (!cfg || !picked_host || !best_index_host || !best_value_host || !workspace || restarts < 1) return GABO_ERR_ARG;
    const int dim = cfg->acq.dim;
    if (dim < 2) return GABO_ERR_DIM;
    const int64_t r = restarts;
    for (int64_t k = 0; k < r; ++k)
        if (picked_host[k] < 0 || picked_host[k] >= max_raw) return GABO_ERR_ARG;
    const gabo::SphSweepWs w = gabo::sph_sweep_layout(workspace, dim, max_raw, r);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (hipMemcpyAsync(w.picked, picked_host, (size_t)r * 8, hipMemcpyHostToDevice, st) != hipSuccess) return GABO_ERR_LAUNCH;
    hipLaunchKernelGGL(gabo::sweep_gather_kernel, dim3((unsigned)((r * dim + 255) / 256)), dim3(256), 0, st, w.raw, w.picked, w.x, r, dim);
    gabo_sphere_acq_params acq = cfg->acq;
    acq.out_sign = -1.0;
    if ((rc = gabo_sphere_acq_eval(w.x, &acq, w.fx, w.eg, r, stream)) != GABO_OK) return rc;
    if ((rc = gabo_sphere_manifold_op(GABO_SPH_PROJ, w.x, w.eg, nullptr, nullptr, w.g, r, dim, stream)) != GABO_OK) return rc;
    hipLaunchKernelGGL(gabo::sweep_rownorm_kernel, dim3((unsigned)((r + 255) / 256)), dim3(256), 0, st, w.g, w.ng, r, dim);
    hipLaunchKernelGGL(gabo::sweep_init_kernel, dim3((unsigned)((r + 255) / 256)), dim3(256), 0, st, w.delta, w.active, w.iters, r, cfg->delta0);
    if (hipMemsetAsync(w.tr, 0, w.tr_bytes, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if ((rc = gabo_sphere_tr_solve(w.x, w.fx, w.g, w.ng, w.delta, w.active, w.iters, &acq, w.tr, w.tr_bytes, r, cfg->theta, cfg->kappa, cfg->mininner,
                                   cfg->maxinner, cfg->exact_hessian, cfg->delta_bar, cfg->rho_prime, cfg->rho_regularization, cfg->mingradnorm,
                                   cfg->maxiter, stream)) != GABO_OK)
        return rc;
    std::vector<double> fx((size_t)r);
    std::vector<int64_t> it((size_t)r);
    if (hipMemcpyAsync(fx.data(), w.fx, (size_t)r * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipMemcpyAsync(it.data(), w.iters, (size_t)r * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    int64_t best = 0, maxit = 0;
    for (int64_t k = 0; k < r; ++k) {
        const double v = -fx[(size_t)k], b = -fx[(size_t)best];
        if ((v > b && b == b) || (v != v && b == b)) best = k;
        if (it[(size_t)k] > maxit) maxit = it[(size_t)k];
    }
    *best_index_host = best;
    *best_value_host = -fx[(size_t)best];
    if (max_iterations_host) *max_iterations_host = maxit;
    if (candidates_dev) *candidates_dev = w.x;
    if (cost_dev) *cost_dev = w.fx;
    if (iterations_dev) *iterations_dev = w.iters;
    return GABO_OK;
}
