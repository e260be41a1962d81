// fp64 matrix pipe vs fp64 vector pipe on gfx950 (MI355X):
//   1. v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 alone (independent accumulators, W waves per SIMD),
//   2. v_fma_f64 alone (same W),
//   3. both at once: every block carries MFMA waves and VALU waves so that each SIMD holds both kinds; the two rates are reported next
//      to what each achieved alone.  Co-issue "works" when the sum of the two exceeds either alone.
//   4. a clock probe: s_memtime (100 MHz constant counter) against s_memrealtime is not available per-SIMD, so the sustained clock is
//      inferred from an integer VALU loop with a known cycle count (v_add_u32 dependent chain: 4 cycles per wave instruction... measured).
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_f64.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

// role 0: MFMA 16x16x4, role 1: MFMA 4x4x4 (4 blocks), role 2: VALU fma.  `mix`: waves 0,1 of a block run role A, waves 2,3 role B
// (a 256-thread block puts one wave on each SIMD; two blocks per CU with opposite parity give every SIMD one wave of each kind).
template <int ACCS>
__device__ __forceinline__ void mfma16_loop(double* out, int iters) {
    v4d acc[ACCS];
#pragma unroll
    for (int c = 0; c < ACCS; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < ACCS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < ACCS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACCS>
__device__ __forceinline__ void mfma4_loop(double* out, int iters) {
    double acc[ACCS];
#pragma unroll
    for (int c = 0; c < ACCS; ++c) acc[c] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < ACCS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < ACCS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__device__ __forceinline__ void valu_loop(double* out, int iters, double a, double b) {
    double acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x * 1e-3 + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// kind: 0 = mfma16 only, 1 = mfma4 only, 2 = valu only, 3 = mfma16 + valu mixed per SIMD, 4 = mfma4 + valu mixed
__global__ __launch_bounds__(512) void bench(double* out, int kind, int it_mfma, int it_valu, double a, double b) {
    const int wave = threadIdx.x >> 6;                 // 8 waves per block: waves w and w+4 share a SIMD
    const bool first = wave < 4;
    if (kind == 0 || (kind == 3 && first)) mfma16_loop<4>(out, it_mfma);
    else if (kind == 1 || (kind == 4 && first)) mfma4_loop<8>(out, it_mfma);
    else valu_loop<8>(out, it_valu, a, b);
}

static float run(double* out, int kind, int blocks, int it_mfma, int it_valu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    bench<<<blocks, 512>>>(out, kind, 10, 10, 1.0000001, 1e-9);
    hipEventRecord(e0);
    bench<<<blocks, 512>>>(out, kind, it_mfma, it_valu, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 4096 * 512 * 8);
    const double simds = 1024.0;
    for (int bpc : {1, 2}) {                            // blocks per CU: 2 or 4 waves per SIMD
        const int blocks = 256 * bpc;
        const double waves = blocks * 8.0;
        // --- alone
        int it16 = 20000, it4 = 20000, itv = 40000;
        float t16 = run(out, 0, blocks, it16, 0);
        double n16 = waves * it16 * 4.0;                // wave-level MFMA instructions
        printf("[%d waves/SIMD] mfma_f64_16x16x4 alone : %.3f ms  %.1f TFLOP/s  %.1f cycles/instr/SIMD @2.4GHz\n", 2 * bpc, t16,
               n16 * 2048.0 / t16 * 1e-9, t16 * 1e-3 * 2.4e9 / (n16 / simds));
        float t4 = run(out, 1, blocks, it4, 0);
        double n4 = waves * it4 * 8.0;
        printf("[%d waves/SIMD] mfma_f64_4x4x4_4b alone: %.3f ms  %.1f TFLOP/s  %.1f cycles/instr/SIMD @2.4GHz\n", 2 * bpc, t4,
               n4 * 512.0 / t4 * 1e-9, t4 * 1e-3 * 2.4e9 / (n4 / simds));
        float tv = run(out, 2, blocks, 0, itv);
        double nv = waves * itv * 8.0;
        printf("[%d waves/SIMD] v_fma_f64 alone        : %.3f ms  %.1f TFLOP/s  %.2f cycles/instr/SIMD @2.4GHz\n", 2 * bpc, tv,
               nv * 128.0 / tv * 1e-9, tv * 1e-3 * 2.4e9 / (nv / simds));
        // --- mixed: half of the waves of every SIMD run MFMA, the other half VALU; iteration counts chosen so that each half alone
        //     would take about the same time
        for (int kind : {3, 4}) {
            double per_mfma_ms = (kind == 3 ? t16 / it16 : t4 / it4) * 2.0;     // alone-time per iteration with half the waves ~ same
            int itm = 10000;
            int itvv = (int)(itm * per_mfma_ms / (tv / itv * 2.0));
            float tm = run(out, kind, blocks, itm, itvv);
            double nm = waves / 2 * itm * (kind == 3 ? 4.0 : 8.0), nvv = waves / 2 * itvv * 8.0;
            double fm = nm * (kind == 3 ? 2048.0 : 512.0), fv = nvv * 128.0;
            // what the same instruction counts would take back to back on one pipe at the alone rates
            double serial_ms = nm / (kind == 3 ? n16 / t16 : n4 / t4) + nvv / (nv / tv);
            printf("[%d waves/SIMD] %s + v_fma_f64 on the same SIMDs: %.3f ms; MFMA %.1f + VALU %.1f = %.1f TFLOP/s; the two alone back to back "
                   "would take %.3f ms -> overlap factor %.2f\n", 2 * bpc, kind == 3 ? "mfma16x16x4" : "mfma4x4x4_4b", tm, fm / tm * 1e-9,
                   fv / tm * 1e-9, (fm + fv) / tm * 1e-9, serial_ms, serial_ms / tm);
        }
    }
    return 0;
}
