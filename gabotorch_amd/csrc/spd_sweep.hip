// Host driver of ONE multi-start acquisition sweep on S^d_++ (config 4 of BASELINE.json: gabo_spd, 512 restarts):
//   joint_optimize_manifold -> gen_batch_initial_conditions_manifold (raw samples drawn and scored, manifold_optimize.py:232-321)
//                           -> gen_candidates_manifold (the restarts' local solves, :124-228) -> get_best_candidates (:118-120).
// Rounds 4-5 enqueued the Python path's launches from C++ (two host calls around the selection heuristic, ten launches in front of the solve kernel
// and three behind it).  Round 6: score the raw samples into a table (sampler + evaluation), select the restarts on the device, start every restart
// from its picked row, solve (ends with the result row) - five launches behind one host wait, plus one host call for the GP's set-up; the tables are laid out so that a multi-GPU
// sweep all_gathers exactly them.  Below, after the helpers: gabo_spd_gp_prepare, gabo_spd_sweep_score_rows / _select_rows / _solve_rows, then the
// sphere twin (two host calls around the host heuristic, as in round 5).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/gabo_hip.h"
#include "spd_tr_body.hpp"
#include "spd_acq_kernel.hpp"
#include "gabo_philox.hpp"

namespace gabo {

int spd_sample_rows(double* out, int64_t out_stride, int64_t first, int64_t n, int d, double min_eig, double max_eig, uint64_t seed, hipStream_t st);   // spd_sample.hip

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

// ---- round 6: the same sweep with its set-up, start and end inside the launches ------------------------------------------------------------------
// gabo_spd_gp_prepare:       Gram of the training set -> Cholesky factor, its inverse, alpha -> entry-major training factors: four launches from ONE host
//                            call (rounds 4-5: ~0.24 ms of Python around them, the GPU waiting for the next launch most of that time)
// gabo_spd_sweep_score_rows: samples [first, first + count) of the stream (or the caller's host draws) as rows [value, Mandel vector] of ONE table, their
//                            acquisition values into the table and, densely, into MAPPED HOST memory: two launches, no copy
// gabo_spd_sweep_solve_rows: every wave starts its restart from its picked row (pre- / post-processing maps, value, gradient, Riemannian gradient and
//                            norm: spd_tr_start_kernel), then iterates to the end (the solve launch) and leaves [cost, iterations, Mandel vector] in a
//                            result row, on the device and in mapped host memory.
// Rows, because a multi-GPU sweep all_gathers exactly two things: the scored raw-sample rows and the result rows - the caller does that on
// the tables themselves between / after the calls (manifold_optimize.py of this package), the driver has no collective of its own.
namespace gabo {

struct RowsWs {
    double *raw_rows, *res_rows, *raw_mat;
    double *x, *fx, *g, *ng, *delta;
    int64_t* iters;
    uint8_t* active;
    void* tr;
    size_t tr_bytes, bytes;
};

static RowsWs rows_layout(void* base, int64_t n, int d, int64_t max_raw, int64_t r, int c) {
    RowsWs w;
    const int64_t dv = (int64_t)d * (d + 1) / 2, dd = (int64_t)d * d;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += (bytes + 255) & ~(size_t)255;
        return q;
    };
    w.raw_rows = (double*)take((size_t)max_raw * (1 + dv) * 8);
    w.res_rows = (double*)take((size_t)r * (2 + dv) * 8);
    w.raw_mat = (double*)take((size_t)max_raw * dd * 8);        // staging of host-drawn matrices
    w.x = (double*)take((size_t)r * dd * 8);
    w.fx = (double*)take((size_t)r * 8);
    w.g = (double*)take((size_t)r * dd * 8);
    w.ng = (double*)take((size_t)r * 8);
    w.delta = (double*)take((size_t)r * 8);
    w.iters = (int64_t*)take((size_t)r * 8);
    w.active = (uint8_t*)take((size_t)r);
    w.tr_bytes = gabo_spd_tr_workspace_bytes(r, d, c, n);
    w.tr = take(w.tr_bytes + 8);
    w.bytes = (size_t)(p - (char*)base);
    return w;
}

// rows [., Mandel vector] from d x d matrices: matrix_to_mandel_kernel's statement (mandel.hip), strided output
__global__ __launch_bounds__(256) void sweep_rows_from_matrices_kernel(const double* __restrict__ mat, double* __restrict__ rows, int64_t n, int d,
                                                                       int64_t row_stride) {
    const int64_t dd = (int64_t)d * d;
    const int dv = d * (d + 1) / 2;
    const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx >= n * dv) return;
    const int64_t q = gidx / dv;
    const int e = (int)(gidx - q * dv);
    int k = 0;
    while (k + 1 < d && (k + 1) * d - (k + 1) * k / 2 <= e) ++k;
    const int c = e - (k * d - k * (k - 1) / 2);
    const int r = c + k;
    const double* m = mat + q * dd;
    rows[q * row_stride + 1 + e] = (k == 0) ? m[r * d + c] : 0.5 * (kSqrt2 * m[c * d + r] + kSqrt2 * m[r * d + c]);
}

// The address the DEVICE uses for memory the host reads or writes directly: device memory as it is; page-locked host memory through its mapping.
// Anything else (pageable host memory) is refused - a kernel that dereferenced it would fault.
static void* device_visible(const void* p) {
    if (!p) return nullptr;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (attr.type == hipMemoryTypeHost) return attr.devicePointer;
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return const_cast<void*>(p);
    return nullptr;
}

}  // namespace gabo

extern "C" size_t gabo_spd_gp_prepare_workspace_bytes(int64_t n, int d) {
    if (n < 1 || d < 2) return 0;
    return (((size_t)n * n * 8 + 255) & ~(size_t)255) + gabo_spd_ai_workspace_bytes(1, n, n, d);
}

extern "C" int gabo_spd_gp_prepare(const double* train_mandel, const double* y, int64_t n, int d, double beta, int flags, double outputscale, double noise,
                                   double mean, double* linv, double* linv_t, double* alpha, double* kinv, double* train_factors, void* workspace,
                                   size_t workspace_bytes, int* status, int* factor_status, gabo_stream_t stream) {
    if (!train_mandel || !y || !linv || !linv_t || !alpha || !workspace || !status || !factor_status || n < 1) return GABO_ERR_ARG;
    if (d < 2 || d > GABO_SPD_REG_MAX_DIM) return GABO_ERR_DIM;
    if (n > GABO_GP_FACTOR_MAX_N) return GABO_ERR_DIM;
    const int out = flags & GABO_OUT_MASK;
    if ((flags & ~GABO_OUT_MASK) || (out != GABO_OUT_GAUSSIAN && out != GABO_OUT_LAPLACE)) return GABO_ERR_ARG;
    if (workspace_bytes < gabo_spd_gp_prepare_workspace_bytes(n, d)) return GABO_ERR_ARG;
    double* kb = (double*)workspace;
    char* pw = (char*)workspace + (((size_t)n * n * 8 + 255) & ~(size_t)255);
    int rc;
    // what SpdAffineInvariant{Gaussian,Laplace}Kernel.forward(X, X) launches for the training set (the x1-is-x2 build), then models.ExactGP's cache
    if ((rc = gabo_spd_ai_pairwise(train_mandel, train_mandel, kb, nullptr, 1, n, n, d, 0, 0, beta, out | GABO_SYMMETRIC, pw,
                                   gabo_spd_ai_workspace_bytes(1, n, n, d), status, stream)) != GABO_OK)
        return rc;
    if ((rc = gabo_gp_factor(kb, y, n, outputscale, noise, mean, linv, linv_t, alpha, kinv, factor_status, stream)) != GABO_OK) return rc;
    if (train_factors && (rc = gabo_spd_acq_prepare_train(train_mandel, train_factors, n, d, status, stream)) != GABO_OK) return rc;
    return GABO_OK;
}

extern "C" size_t gabo_spd_sweep_rows_workspace_bytes(int64_t n_train, int d, int64_t max_raw, int64_t restarts, int n_constraints) {
    if (n_train < 1 || d < 2 || max_raw < 0 || restarts < 0 || n_constraints < 0) return 0;
    return gabo::rows_layout(nullptr, n_train, d, max_raw, restarts, n_constraints).bytes;
}

extern "C" int gabo_spd_sweep_rows_tables(void* workspace, int64_t n_train, int d, int64_t max_raw, int64_t restarts, int n_constraints, double** raw_rows,
                                          double** result_rows) {
    if (!workspace || n_train < 1 || d < 2 || max_raw < 0 || restarts < 0 || n_constraints < 0) return GABO_ERR_ARG;
    const gabo::RowsWs w = gabo::rows_layout(workspace, n_train, d, max_raw, restarts, n_constraints);
    if (raw_rows) *raw_rows = w.raw_rows;
    if (result_rows) *result_rows = w.res_rows;
    return GABO_OK;
}

extern "C" int gabo_spd_sweep_score_rows(const gabo_spd_sweep_config* cfg, int64_t first_sample, int64_t first_row, int64_t count, int64_t max_raw,
                                         int64_t restarts, uint64_t seed, const double* raw_matrices_host, double* values_mapped, void* workspace,
                                         size_t workspace_bytes, int* status, int* status_mapped, int synchronize, gabo_stream_t stream) {
    if (!cfg || !workspace || !status || first_sample < 0 || first_row < 0 || count < 1 || first_row + count > max_raw || restarts < 1) return GABO_ERR_ARG;
    const int d = cfg->d;
    if (d < 2 || d > 8 || cfg->n_constraints < 0 || cfg->n_constraints > GABO_SWEEP_MAX_CONSTRAINTS) return GABO_ERR_DIM;
    const gabo_spd_acq_params& a = cfg->acq;
    if (a.n < 1 || a.n > gabo_spd_acq_max_train(d) || !a.train_factors || !a.alpha) return GABO_ERR_ARG;
    if (a.kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!a.linv || !a.linv_t)) return GABO_ERR_ARG;
    const gabo::RowsWs w = gabo::rows_layout(workspace, a.n, d, max_raw, restarts, cfg->n_constraints);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    double* mirror = nullptr;
    if (values_mapped && !(mirror = (double*)gabo::device_visible(values_mapped))) return GABO_ERR_ARG;
    int* smirror = nullptr;
    if (status_mapped && !(smirror = (int*)gabo::device_visible(status_mapped))) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t dv = (int64_t)d * (d + 1) / 2, stride = 1 + dv;
    double* rows = w.raw_rows + first_row * stride;
    int rc;
    if (raw_matrices_host) {
        // the caller's own sampler (`manifold.rand` is user code: the reference binds spd_sample to it, examples/gabo_spd.py:102) drew them on the host
        if (hipMemcpyAsync(w.raw_mat, raw_matrices_host, (size_t)count * d * d * 8, hipMemcpyHostToDevice, st) != hipSuccess) return GABO_ERR_LAUNCH;
        hipLaunchKernelGGL(gabo::sweep_rows_from_matrices_kernel, dim3((unsigned)((count * dv + 255) / 256)), dim3(256), 0, st, w.raw_mat, rows, count, d,
                           stride);
        if (hipGetLastError() != hipSuccess) return GABO_ERR_LAUNCH;
    } else {
        if (!(cfg->min_eig > 0.0) || !(cfg->max_eig >= cfg->min_eig)) return GABO_ERR_ARG;
        // (Mandel output of the sampler: kSqrt2 * s, the bits matrix_to_mandel gives for the symmetric matrix it would have written)
        if ((rc = gabo::spd_sample_rows(rows + 1, stride, first_sample, count, d, cfg->min_eig, cfg->max_eig, seed, st)) != GABO_OK) return rc;
    }
    gabo_spd_acq_params acq = a;
    acq.out_sign = 1.0;              // (`values` are the acquisition values themselves)
    gabo::AcqLaunch al{rows + 1, acq, rows, nullptr, nullptr, count, d, nullptr, status, st};
    al.x_stride = stride;
    al.value_stride = stride;
    al.value_mirror = mirror;
    al.status_mirror = smirror;
    if ((rc = gabo::acq_launch(al)) != GABO_OK) return rc;
    if (synchronize && hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    return GABO_OK;
}

namespace gabo {

// ---- which raw samples become restarts, on the device -------------------------------------------------------------------------------------------
// gen_batch_initial_conditions_manifold hands the scored raw samples to botorch's initialize_q_batch_nonneg ([3P] botorch.optim.initializers;
// manifold_optimize.py:296-317 of the reference; this package's restatement: models.initialize_q_batch_nonneg / select_rows): the samples whose
// value is at least alpha * max (alpha shrunk by tens until n of them qualify) are drawn WITHOUT REPLACEMENT with weights
// exp(eta (y / max - 1)), and the arg-max is forced into the last slot when the draw missed it.  torch.multinomial draws without replacement by
// the exponential race - keys w_i / E_i, E_i ~ Exp(1), the n largest keys win - and so does this kernel, with E_i = -log(u_i) from the Philox
// stream (seed, sample index): the same distribution, the library's own random stream instead of the host generator's.
// Done on the host (rounds 3-5, and still the path for acquisition functions that are not non-negative, for more than kSelectMaxTotal raw samples
// and when this kernel reports that the heuristic has to fall back) the step costs a device -> host wait, 0.1-0.25 ms of tensor bookkeeping and
// a host -> device hand-over in the MIDDLE of a 1.2-ms sweep; here it is one launch of one workgroup between the scoring and the solve launches.
// One workgroup of 1024 threads; keys and sample indices live in LDS (total <= kSelectMaxTotal); the n largest keys, in order, come from a bitonic
// sort (ties: the lower sample index).  (Counting, for every key, the keys that beat it - total^2 / 1024 comparisons per thread - was the first
// version: fine at 256 raw samples, ~0.2 ms at 2048.)
constexpr int kSelectMaxTotal = 8192;

// block-wide reduction, every thread gets the result: shuffles inside a wave, one LDS slot per wave, ONE barrier (the scratch has two banks used
// in turn: a wave can only reach the reduction after next once every wave has read this one's bank)
template <typename T, typename Op>
static __device__ __forceinline__ T block_reduce_1024(T v, T* scratch, int& bank, Op op) {
    for (int o = 32; o >= 1; o >>= 1) v = op(v, __shfl_xor(v, o));
    T* sl = scratch + 16 * bank;
    bank ^= 1;
    if ((threadIdx.x & 63) == 0) sl[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = sl[0];
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) r = op(r, sl[wv]);
    return r;
}

// flag: 0 = picked; 1 = the heuristic needs its random fall-backs (no positive value, fewer positive values than restarts) or the values hold a NaN:
// the caller selects on the host.  picked_rows: the table rows of THIS rank's restarts (restart k belongs to rank k % world) in restart order;
// picked_samples (may be null): the sample index of every restart.
__global__ __launch_bounds__(1024) void sweep_select_kernel(const double* __restrict__ raw_rows, int64_t stride, int total, int per, int n, double eta,
                                                            double alpha, uint64_t seed, int seed_in_header, int rank, int world,
                                                            int64_t* __restrict__ picked_rows, int64_t* __restrict__ picked_samples,
                                                            int* __restrict__ flag_dev, int* __restrict__ flag_mapped) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int P2 = 1;
    while (P2 < total) P2 <<= 1;
    double* key = sm;                      // P2 (total rounded up to a power of two)
    double* dred = sm + P2;                // 1024
    int* ired = (int*)(dred + 1024);       // 1024
    int* sidx_of = ired + 1024;            // P2
    __shared__ int max_rank;
    const int tid = threadIdx.x;
    auto row_of = [&](int smp) -> int64_t { return (int64_t)(smp / per) * (per + 1) + 1 + smp % per; };
    if (seed_in_header) seed = (uint64_t)raw_rows[0];       // (rank 0's proposal: the header row of block 0, < 2^52)
    double best = -__builtin_inf();
    int npos = 0, nnan = 0;
    for (int sidx = tid; sidx < total; sidx += 1024) {
        const double v = raw_rows[row_of(sidx) * stride];
        key[sidx] = v;
        nnan += (v != v) ? 1 : 0;
        npos += (v > 0.0) ? 1 : 0;
        best = v > best ? v : best;
    }
    int dbank = 0, ibank = 0;
    const double max_val = block_reduce_1024<double>(best, dred, dbank, [](double a, double b) { return a > b ? a : b; });
    int first = 0x7fffffff;
    for (int sidx = tid; sidx < total; sidx += 1024)
        if (key[sidx] == max_val && sidx < first) first = sidx;
    const int max_idx = block_reduce_1024<int>(first, ired, ibank, [](int a, int b) { return a < b ? a : b; });
    const int num_pos = block_reduce_1024<int>(npos, ired, ibank, [](int a, int b) { return a + b; });
    const int num_nan = block_reduce_1024<int>(nnan, ired, ibank, [](int a, int b) { return a + b; });
    if (num_nan > 0 || !(max_val > 0.0) || num_pos < n) {
        if (tid == 0) {
            *flag_dev = 1;
            if (flag_mapped) *flag_mapped = 1;
        }
        return;
    }
    double thr = alpha * max_val;
    for (int guard = 0; guard < 400; ++guard) {           // (alpha -> 0: thr -> 0 and the num_pos >= n positive values all qualify)
        int c = 0;
        for (int sidx = tid; sidx < total; sidx += 1024) c += (key[sidx] >= thr) ? 1 : 0;
        const int cnt = block_reduce_1024<int>(c, ired, ibank, [](int a, int b) { return a + b; });
        if (cnt >= n) break;
        alpha = 0.1 * alpha;
        thr = alpha * max_val;
    }
    for (int sidx = tid; sidx < total; sidx += 1024) {
        const double v = key[sidx];
        Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint64_t)sidx, 0u, 0x73656c65u};
        double u1, u2;
        rng.uniform2(u1, u2);
        const double w = exp(eta * (v / max_val - 1.0));
        key[sidx] = (v >= thr) ? w / -log(u1) : -1.0;      // (u1 in (0, 1]: E >= 0; E = 0 gives +inf, the sure winner it should be)
    }
    // the n largest keys, in order: bitonic sort of (key, sample) over the next power of two (padding: key -2, behind every real entry);
    // "a before b" = larger key, ties by the lower sample index
    int P = 1;
    while (P < total) P <<= 1;
    for (int e = tid; e < P; e += 1024) {
        sidx_of[e] = e;
        if (e >= total) key[e] = -2.0;
    }
    if (tid == 0) max_rank = n;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;
                const double ka = key[a], kb = key[b];
                const int ia = sidx_of[a], ib = sidx_of[b];
                const bool b_first = kb > ka || (kb == ka && ib < ia);
                const bool up = (a & k) == 0;
                if (b_first == up) {
                    key[a] = kb; key[b] = ka;
                    sidx_of[a] = ib; sidx_of[b] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < n; r += 1024) {
        const int smp = sidx_of[r];
        if (smp == max_idx) max_rank = r;
        if (picked_samples) picked_samples[r] = smp;
        if (r % world == rank) picked_rows[r / world] = row_of(smp);
    }
    __syncthreads();
    if (tid == 0) {
        if (max_rank >= n) {                               // the draw missed the arg-max: it takes the last slot (botorch: idcs[-1] = max_idx)
            if (picked_samples) picked_samples[n - 1] = max_idx;
            if ((n - 1) % world == rank) picked_rows[(n - 1) / world] = row_of(max_idx);
        }
        *flag_dev = 0;
        if (flag_mapped) *flag_mapped = 0;
    }
}

}  // namespace gabo

extern "C" int gabo_spd_sweep_select_supported(int64_t total, int64_t restarts) {
    return (total >= 1 && total <= gabo::kSelectMaxTotal && restarts >= 1 && restarts < total) ? 1 : 0;
}

extern "C" int gabo_spd_sweep_select_rows(const double* raw_rows, int d, int64_t total, int64_t per_rank, int64_t restarts, double eta, double alpha,
                                          uint64_t seed, int seed_in_header, int rank, int world, int64_t* picked_rows, int64_t* picked_samples,
                                          int* flag, int* flag_mapped, gabo_stream_t stream) {
    if (!raw_rows || !picked_rows || !flag || d < 2 || rank < 0 || world < 1 || rank >= world || per_rank < 1 || per_rank * world < total) return GABO_ERR_ARG;
    if (!gabo_spd_sweep_select_supported(total, restarts)) return GABO_ERR_DIM;
    int* fm = nullptr;
    if (flag_mapped && !(fm = (int*)gabo::device_visible(flag_mapped))) return GABO_ERR_ARG;
    size_t p2 = 1;
    while ((int64_t)p2 < total) p2 <<= 1;
    const size_t lds = p2 * 12 + 1024 * 12;
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return GABO_ERR_LAUNCH;
    if (!(attr_set.load(std::memory_order_acquire) >> dev & 1)) {
        if (hipFuncSetAttribute((const void*)gabo::sweep_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(gabo::kSelectMaxTotal * 12 + 1024 * 12)) != hipSuccess)
            return GABO_ERR_LAUNCH;
        attr_set.fetch_or((uint64_t)1 << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(gabo::sweep_select_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, raw_rows, (int64_t)(1 + d * (d + 1) / 2), (int)total,
                       (int)per_rank, (int)restarts, eta, alpha, seed, seed_in_header, rank, world, picked_rows, picked_samples, flag, fm);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

extern "C" int gabo_spd_sweep_solve_rows(const gabo_spd_sweep_config* cfg, const int64_t* picked_mapped, int64_t restarts, int64_t max_raw,
                                         double* results_mapped, const int* skip_flag, void* workspace, size_t workspace_bytes, int* status,
                                         int* status_mapped, int synchronize, gabo_stream_t stream) {
    if (!cfg || !picked_mapped || !workspace || !status || restarts < 1 || restarts > 0x7fffffffLL) return GABO_ERR_ARG;
    const int d = cfg->d, c = cfg->n_constraints;
    if (d < 2 || d > 8 || c < 0 || c > GABO_SWEEP_MAX_CONSTRAINTS) return GABO_ERR_DIM;
    const gabo_spd_acq_params& a = cfg->acq;
    if (a.n < 1 || a.n > gabo_spd_acq_max_train(d) || !a.train_factors || !a.alpha || cfg->maxinner < 1 || cfg->maxiter < 1) return GABO_ERR_ARG;
    if (a.kind == GABO_ACQ_EXPECTED_IMPROVEMENT && (!a.linv || !a.linv_t)) return GABO_ERR_ARG;
    if (!gabo_spd_tr_solve_supported(&a, restarts, d, c, 0)) return GABO_ERR_DIM;
    const int64_t r = restarts;
    const gabo::RowsWs w = gabo::rows_layout(workspace, a.n, d, max_raw, r, c);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    const int64_t* picked = (const int64_t*)gabo::device_visible(picked_mapped);
    if (!picked) return GABO_ERR_ARG;
    double* res_host = nullptr;
    if (results_mapped && !(res_host = (double*)gabo::device_visible(results_mapped))) return GABO_ERR_ARG;
    int* smirror = nullptr;
    if (status_mapped && !(smirror = (int*)gabo::device_visible(status_mapped))) return GABO_ERR_ARG;
    {
        // (the indices are read here only when the host can: mapped host memory; device-resident indices are the caller's responsibility)
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, picked_mapped) == hipSuccess && attr.type == hipMemoryTypeHost) {
            for (int64_t k = 0; k < r; ++k)
                if (picked_mapped[k] < 0 || picked_mapped[k] >= max_raw) return GABO_ERR_ARG;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    gabo_spd_acq_params acq = a;
    acq.out_sign = -1.0;             // cost = -acquisition
    gabo::BuiltinCons B;
    B.n = c;
    B.strict = cfg->strict ? 1 : 0;
    B.big_dim = 0;
    B.lift_w = B.lift_p = B.lift_x0 = nullptr;
    for (int k = 0; k < gabo::kMaxCons; ++k) {
        B.kind[k] = k < c ? cfg->constraint_kind[k] : 0;
        B.bound[k] = k < c ? cfg->constraint_bound[k] : 0.0;
        if (k < c && B.kind[k] != GABO_CONSTRAINT_MAX_EIGENVALUE && B.kind[k] != GABO_CONSTRAINT_MIN_EIGENVALUE) return GABO_ERR_ARG;
    }
    if (gabo::tr_solve_uses_global_workspace(acq, r, d, c, 0) && hipMemsetAsync(w.tr, 0, w.tr_bytes, st) != hipSuccess) return GABO_ERR_LAUNCH;
    gabo::SolveArgs sa{w.x, w.fx, w.g, w.ng, w.delta, w.active, w.iters, &acq, B, w.tr, r, d, cfg->delta_cons, cfg->theta, cfg->kappa, cfg->mininner,
                       cfg->maxinner, cfg->delta_bar, cfg->rho_prime, cfg->rho_regularization, cfg->mingradnorm, cfg->maxiter, status, st};
    sa.start = gabo::TrStart{w.raw_rows, 1 + (int64_t)d * (d + 1) / 2, picked, cfg->delta0, w.res_rows, res_host, smirror, skip_flag};
    int rc;
    if ((rc = gabo::tr_solve_dispatch(sa)) != GABO_OK) return rc;
    if (synchronize && hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    return GABO_OK;
}

// ---- the sphere twin: gen_batch_initial_conditions_manifold + gen_candidates_manifold + get_best_candidates on S^(dim-1) ----------------------------
// The reference's gabo_sphere examples (examples/bo_sphere/benchmark_examples/gabo_sphere.py:151-175): stock TrustRegions, no constraints, exact or
// finite-difference Hessian-vector products.  Same two-call shape as above; the raw samples always come from the caller's host sampler (`manifold.rand`).
namespace gabo {

struct SphSweepWs {
    double *raw, *raw_val, *x, *fx, *eg, *g, *ng, *delta;
    int64_t *picked, *picked_rows, *iters;
    int* flag;
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
    w.raw_val = (double*)take((size_t)(max_raw + 1) * 8) + 1;      // (one slot in front: the selection kernel addresses sample k as row 1 + k of a table)
    w.x = (double*)take((size_t)r * dim * 8);
    w.fx = (double*)take((size_t)r * 8);
    w.eg = (double*)take((size_t)r * dim * 8);
    w.g = (double*)take((size_t)r * dim * 8);
    w.ng = (double*)take((size_t)r * 8);
    w.delta = (double*)take((size_t)r * 8);
    w.picked = (int64_t*)take((size_t)r * 8);
    w.picked_rows = (int64_t*)take((size_t)r * 8);
    w.iters = (int64_t*)take((size_t)r * 8);
    w.flag = (int*)take(8);
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

// count points uniform on S^(dim-1): normal deviates from the library's Philox stream (seed, sample index), normalised - the distribution of
// pymanopt's Sphere.rand ([3P]: randn then / norm).  One thread per point.
__global__ __launch_bounds__(256) void sphere_sample_kernel(double* __restrict__ out, int64_t count, int dim, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32), (uint64_t)i, 0u};
    ph.tag = 0x73706872u;      // "sphr": this sampler's stream
    double* row = out + i * dim;
    double ss = 0.0;
    for (int k = 0; k < dim; k += 2) {
        double z0, z1;
        ph.normal2(z0, z1);
        row[k] = z0;
        ss = __builtin_fma(z0, z0, ss);
        if (k + 1 < dim) {
            row[k + 1] = z1;
            ss = __builtin_fma(z1, z1, ss);
        }
    }
    const double inv = 1.0 / __builtin_sqrt(ss);
    for (int k = 0; k < dim; ++k) row[k] *= inv;
}

// dst row i = src row idx[i], the index clamped into the table (when the selection kernel raised its fall-back flag it picked nothing: the launches behind
// it then work on whatever rows these are, and the caller discards the result)
__global__ __launch_bounds__(256) void sweep_gather_clamped_kernel(const double* __restrict__ src, const int64_t* __restrict__ idx, double* __restrict__ dst,
                                                                  int64_t r, int dim, int64_t rows) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= r * dim) return;
    int64_t k = idx[e / dim];
    k = k < 0 ? 0 : (k >= rows ? rows - 1 : k);
    dst[e] = src[k * dim + e % dim];
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

// The same sweep in ONE host call with one wait: raw samples (the caller's, or drawn here on the library's Philox stream when raw_points_host is null) ->
// acquisition values -> restart selection on the device (sweep_select_kernel, the SPD sweep's) -> start of every restart -> the solve -> final costs and
// iteration counts to the host, arg-max.  *fallback_host = 1: the selection kernel reported that botorch's heuristic needs its random fall-backs (or the
// values hold a NaN); nothing else was written and the caller runs gabo_sphere_sweep_score / host selection / gabo_sphere_sweep_solve.
extern "C" int gabo_sphere_sweep_run(const gabo_sphere_sweep_config* cfg, int64_t count, int64_t restarts, const double* raw_points_host, uint64_t sample_seed,
                                     double eta, double alpha, uint64_t select_seed, int64_t* best_index_host, double* best_value_host,
                                     int64_t* max_iterations_host, double** candidates_dev, double** cost_dev, int64_t** iterations_dev, int64_t** picked_dev,
                                     int* fallback_host,
                                     void* workspace, size_t workspace_bytes, gabo_stream_t stream) {
    if (!cfg || !best_index_host || !best_value_host || !fallback_host || !workspace || count < 1 || restarts < 1) return GABO_ERR_ARG;
    if (!gabo_spd_sweep_select_supported(count, restarts)) return GABO_ERR_DIM;
    const int dim = cfg->acq.dim;
    if (dim < 2) return GABO_ERR_DIM;
    const int64_t r = restarts;
    const gabo::SphSweepWs w = gabo::sph_sweep_layout(workspace, dim, count, r);
    if (w.bytes > workspace_bytes) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (raw_points_host) {
        if (hipMemcpyAsync(w.raw, raw_points_host, (size_t)count * dim * 8, hipMemcpyHostToDevice, st) != hipSuccess) return GABO_ERR_LAUNCH;
    } else {
        hipLaunchKernelGGL(gabo::sphere_sample_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, w.raw, count, dim, sample_seed);
    }
    gabo_sphere_acq_params acq = cfg->acq;
    acq.out_sign = 1.0;
    if ((rc = gabo_sphere_acq_eval(w.raw, &acq, w.raw_val, nullptr, count, stream)) != GABO_OK) return rc;
    {   // selection: the values as a one-column table whose row 1 + k is sample k (sweep_select_kernel's addressing with one block of `count` rows)
        size_t p2 = 1;
        while ((int64_t)p2 < count) p2 <<= 1;
        const size_t lds = p2 * 12 + 1024 * 12;
        static std::atomic<uint64_t> attr_set{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return GABO_ERR_LAUNCH;
        if (!(attr_set.load(std::memory_order_acquire) >> dev & 1)) {
            if (hipFuncSetAttribute((const void*)gabo::sweep_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(gabo::kSelectMaxTotal * 12 + 1024 * 12)) != hipSuccess)
                return GABO_ERR_LAUNCH;
            attr_set.fetch_or((uint64_t)1 << dev, std::memory_order_release);
        }
        hipLaunchKernelGGL(gabo::sweep_select_kernel, dim3(1), dim3(1024), lds, st, w.raw_val - 1, (int64_t)1, (int)count, (int)count, (int)r, eta, alpha,
                           select_seed, 0, 0, 1, w.picked_rows, w.picked, w.flag, (int*)nullptr);
    }
    hipLaunchKernelGGL(gabo::sweep_gather_clamped_kernel, dim3((unsigned)((r * dim + 255) / 256)), dim3(256), 0, st, w.raw, w.picked, w.x, r, dim, count);
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
    int flag = 0;
    if (hipMemcpyAsync(fx.data(), w.fx, (size_t)r * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipMemcpyAsync(it.data(), w.iters, (size_t)r * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipMemcpyAsync(&flag, w.flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return GABO_ERR_LAUNCH;
    *fallback_host = flag != 0 ? 1 : 0;
    if (flag != 0) return GABO_OK;
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
    if (picked_dev) *picked_dev = w.picked;          // the raw-sample index of every restart
    return GABO_OK;
}
