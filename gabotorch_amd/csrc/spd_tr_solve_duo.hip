// Single-launch trust-region solve, affine-invariant surrogate, two waves per restart (templates and description in spd_tr_duo_body.hpp).
// In this translation unit __syncthreads() is the WAVE-level fence it already is in the 64-thread kernels the shared bodies were written for; the barriers
// between the two waves of a block are explicit (duo_block_sync).
#include <hip/hip_runtime.h>

#include <cstdlib>

namespace gabo {
static __device__ __forceinline__ void duo_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace gabo
#define __syncthreads() ::gabo::duo_wave_sync()
#ifdef GABO_DUO_TIMES       /* development (tools/duo_times.py): lane 0 of both waves of one restart records (tag + 1000 wave, cycle) pairs */
#ifndef GABO_DUO_CLOCKS_BLOCK
#define GABO_DUO_CLOCKS_BLOCK 55
#endif
static __device__ long long gabo_duo_clk[2 * 8192];
static __device__ int gabo_duo_clk_n;
#define GABO_TICK(tag)                                                                        \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x == GABO_DUO_CLOCKS_BLOCK) {                        \
            const int k_ = atomicAdd(&gabo_duo_clk_n, 1);                                     \
            if (k_ < 8192) {                                                                  \
                gabo_duo_clk[2 * k_] = (tag) + 1000 * (int)threadIdx.y;                       \
                gabo_duo_clk[2 * k_ + 1] = (long long)__builtin_amdgcn_s_memtime();           \
            }                                                                                 \
        }                                                                                     \
    } while (0)
#endif
#define GABO_DUO_TU 1
#include "spd_tr_duo_body.hpp"

#ifndef GABO_DUO_MIN_DIM
#define GABO_DUO_MIN_DIM 2
#endif
#ifndef GABO_DUO_MAX_DIM
#define GABO_DUO_MAX_DIM 6
#endif
#ifndef GABO_DUO_MAX_RESTARTS
#define GABO_DUO_MAX_RESTARTS 512      /* 2 waves x 512 blocks = one wave on each of the 1024 SIMDs */
#endif

namespace gabo {

static __device__ int g_duo_counters[2];

static size_t duo_static_lds(int d) {
    switch (d) {
        case 2: return sizeof(DuoStatic<2>);
        case 3: return sizeof(DuoStatic<3>);
        case 4: return sizeof(DuoStatic<4>);
        case 5: return sizeof(DuoStatic<5>);
        case 6: return sizeof(DuoStatic<6>);
        case 7: return sizeof(DuoStatic<7>);
        case 8: return sizeof(DuoStatic<8>);
    }
    return (size_t)1 << 30;
}

// whether solve_affine_invariant_duo takes this problem: the latency regime of the one-wave kernel (everything in LDS), the symmetric inverse of the Gram
// matrix, eigenvalue bounds of the iterate itself (kinds 0 / 1) or no constraints.  GABO_TR_DUO=0 in the environment keeps the one-wave kernel (A/B, tests).
static int& duo_enabled() {
    static int enabled = []() { const char* e = getenv("GABO_TR_DUO"); return (e && e[0] == '0') ? 0 : 1; }();
    return enabled;
}
bool solve_duo_wanted(const SolveArgs& a) {
    if (!duo_enabled()) return false;
    if (a.d < GABO_DUO_MIN_DIM || a.d > GABO_DUO_MAX_DIM || a.r > GABO_DUO_MAX_RESTARTS) return false;
    if ((a.P->flags & GABO_METRIC_MASK) != GABO_METRIC_AFFINE_INVARIANT) return false;
    if (!a.P->linv || a.P->linv != a.P->linv_t) return false;
    if (builtin_has_kind(a.B, true)) return false;
    const size_t total = duo_lds_layout(a.P->n, a.d, a.B.n).bytes + duo_static_lds(a.d);
    return total <= 64 * 1024;
}

template <int D>
static int launch_duo(const SolveArgs& a) {
    if (a.start.raw_rows != nullptr) {
        int sgp = 0;
        const size_t slds = tr_dynamic_lds(a.P->n, a.r, &sgp, tr_factor_count(*a.P));
        hipLaunchKernelGGL((spd_tr_start_kernel<D, 0>), dim3((unsigned)a.r), dim3(64), slds, a.st, a.x, a.fx, a.g, a.ng, a.delta_tr, a.active, a.iters,
                           *a.P, a.ws, a.r, a.B.n, a.status, sgp, a.start);
    }
    int* counters = nullptr;
    if (hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_duo_counters)) != hipSuccess) counters = nullptr;
    const size_t lds = duo_lds_layout(a.P->n, D, a.B.n).bytes;
    hipLaunchKernelGGL((spd_tr_solve_duo_kernel<D>), dim3((unsigned)a.r), dim3(64, 2), lds, a.st, a.x, a.fx, a.g, a.ng, a.delta_tr, a.active, a.iters,
                       *a.P, a.B, a.r, a.delta_cons, a.theta, a.kappa, a.mininner, a.maxinner, a.delta_bar, a.rho_prime, a.rho_regularization,
                       a.mingradnorm, a.maxiter, a.status, a.shortcuts, a.rec, a.rec_cap, a.start, counters);
    return hipGetLastError() == hipSuccess ? GABO_OK : GABO_ERR_LAUNCH;
}

int solve_affine_invariant_duo(const SolveArgs& a) {
    switch (a.d) {
#define GABO_CASE(DD)                                                              \
    case DD:                                                                       \
        if constexpr (DD >= GABO_DUO_MIN_DIM && DD <= GABO_DUO_MAX_DIM) return launch_duo<DD>(a); \
        else return GABO_ERR_DIM;
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8)
#undef GABO_CASE
    }
    return GABO_ERR_DIM;
}

}  // namespace gabo

extern "C" int gabo_spd_tr_two_waves(int enable) {
    const int before = gabo::duo_enabled();
    if (enable == 0 || enable == 1) gabo::duo_enabled() = enable;
    return before;
}

extern "C" int gabo_spd_tr_two_waves_counters(long long* hits, long long* misses, int reset) {
    int h[2] = {0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gabo::g_duo_counters), sizeof(h)) != hipSuccess) return GABO_ERR_LAUNCH;
    if (hits) *hits = h[0];
    if (misses) *misses = h[1];
    if (reset) {
        const int z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gabo::g_duo_counters), z, sizeof(z)) != hipSuccess) return GABO_ERR_LAUNCH;
    }
    return GABO_OK;
}

#ifdef GABO_DUO_TIMES
extern "C" int gabo_debug_duo_clocks(long long* out, int max_pairs, int reset) {
    int n = 0;
    hipMemcpyFromSymbol(&n, HIP_SYMBOL(gabo_duo_clk_n), sizeof(int));
    if (n > max_pairs) n = max_pairs;
    if (n > 8192) n = 8192;
    hipMemcpyFromSymbol(out, HIP_SYMBOL(gabo_duo_clk), (size_t)n * 2 * sizeof(long long));
    if (reset) {
        const int zero = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(gabo_duo_clk_n), &zero, sizeof(int));
    }
    return n;
}
extern "C" int gabo_debug_duo_times(long long* out, int restarts) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gabo::g_duo_times), (size_t)restarts * 4 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
