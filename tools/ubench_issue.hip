// Issue cost per instruction KIND on gfx950 (MI355X), in SHADER cycles measured inside the kernel:
//   every wave brackets its loop with s_memtime (shader clock) and s_memrealtime (100 MHz constant clock), so the table gives both the
//   cycles per wave-instruction per SIMD and the clock the chip actually ran at under that load (a power-limited fp64 clock would show
//   here and nowhere else).  8 independent chains per lane, W waves per SIMD, the whole chip busy (256 CUs x W blocks of 4 waves).
// Kinds: v_fma_f64 with 3 / 2 / 1 vector operands, v_mul_f64, v_add_f64, v_max_f64, v_rcp_f64, v_rsq_f64, v_cndmask pair, v_cmp_f64,
//   v_fma_f32, v_pk_fma_f32, v_mov_b64, v_ldexp_f64, v_frexp_mant_f64, and the MFMA f64 for reference.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { FMA3, FMA2S, FMA1SC, FMA_DISTINCT, MUL, ADD, MAXK, RCP, RSQ, CNDMASK2, CMP, FMA32, PKFMA32, MOV64, LDEXP, FREXP, MIX_QL, NKINDS };
static const char* kNames[NKINDS] = {"v_fma_f64 d,d,v,v (3 VGPR sources)", "v_fma_f64 d,d,v,s (2 VGPR + SGPR)", "v_fma_f64 d,d,s,1.0 (1 VGPR + SGPR + const)",
                                     "v_fma_f64 d,x,y,z (dest not a source)", "v_mul_f64 d,d,v", "v_add_f64 d,d,v", "v_max_f64 d,d,v", "v_rcp_f64", "v_rsq_f64",
                                     "2 x v_cndmask_b32 (one f64 select)", "v_cmp_lt_f64 (vcc)", "v_fma_f32", "v_pk_fma_f32", "v_mov_b64",
                                     "v_ldexp_f64", "v_frexp_mant_f64", "QL-step-like mix (10 fma/mul/add + 1 rcp + 1 max)"};
static const int kInstrPerChainStep[NKINDS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 12};

template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(double* out, uint64_t* clocks, int iters, double a_in, double b_in) {
    double acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = 1.0 + threadIdx.x * 1e-3 + c * 1e-2;
    double a = a_in + threadIdx.x * 1e-12, b = b_in + threadIdx.x * 1e-13;   // VGPR operands
    double x = a * 1.5, y = b * 2.5;
    float fa = (float)a, fb = (float)b;
    float facc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) facc[c] = (float)acc[c];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pacc[CHAINS], pa = {fa, fa}, pb = {fb, fb};
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) pacc[c] = v2f{facc[c], facc[c]};
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if constexpr (KIND == FMA3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
            else if constexpr (KIND == FMA2S) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "s"(b_in));
            else if constexpr (KIND == FMA1SC) asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(acc[c]) : "s"(a_in));
            else if constexpr (KIND == FMA_DISTINCT) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(acc[c]) : "v"(a), "v"(b), "v"(x));
            else if constexpr (KIND == MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(a));
            else if constexpr (KIND == ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
            else if constexpr (KIND == MAXK) asm volatile("v_max_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
            else if constexpr (KIND == RCP) asm volatile("v_rcp_f64 %0, %0" : "+v"(acc[c]));
            else if constexpr (KIND == RSQ) asm volatile("v_rsq_f64 %0, %0" : "+v"(acc[c]));
            else if constexpr (KIND == CNDMASK2) {
                uint32_t lo = __double2loint(acc[c]), hi = __double2hiint(acc[c]);
                asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(__double2loint(a)), "v"(__double2hiint(a)) : );
                acc[c] = __hiloint2double(hi, lo);
            } else if constexpr (KIND == CMP) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(acc[c]), "v"(a) : "vcc");
            else if constexpr (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(facc[c]) : "v"(fa), "v"(fb));
            else if constexpr (KIND == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pacc[c]) : "v"(pa), "v"(pb));
            else if constexpr (KIND == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(acc[c]) : "v"(a));
            else if constexpr (KIND == LDEXP) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(acc[c]));
            else if constexpr (KIND == FREXP) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(acc[c]));
            else if constexpr (KIND == MIX_QL) {
                // the shape of one root-free QL sweep step: a dependent chain of mul/fma with one reciprocal and one max
                double p = acc[c], r, t, f;
                asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(p), "v"(b));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(p), "v"(r));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(t));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(r) : "v"(t), "v"(a));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(t) : "v"(r), "v"(b));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(f) : "v"(t), "v"(p));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f) : "v"(a));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(r) : "v"(f));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f) : "v"(p), "v"(r));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(f));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(f), "v"(t));
                asm volatile("v_max_f64 %0, %0, %1" : "+v"(p) : "v"(b));
                acc[c] = p;
                asm volatile("" : "+v"(r));
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t r1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c] + (double)facc[c] + (double)pacc[c][0];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) {
        const int wave = gid >> 6;
        clocks[2 * wave] = t1 - t0;
        clocks[2 * wave + 1] = r1 - r0;
    }
}

template <int KIND>
static void run(double* out, uint64_t* clocks, int wps) {
    const int blocks = 256 * wps, waves = blocks * 4;
    const int iters = KIND == MIX_QL ? 3000 : (KIND == RCP || KIND == RSQ ? 8000 : 20000);
    issue_kernel<KIND><<<blocks, 256>>>(out, clocks, 200, 1.0000001, 1e-9);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    issue_kernel<KIND><<<blocks, 256>>>(out, clocks, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * waves);
    hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; real += (double)h[2 * w + 1]; }
    cyc /= waves;
    real /= waves;                                              // 100 MHz ticks
    const double instr_per_wave = (double)iters * CHAINS * kInstrPerChainStep[KIND];
    const double mhz = cyc / (real / 100.0);                    // shader MHz while the loop ran
    // per SIMD: wps waves share it, so the SIMD issued wps * instr_per_wave instructions in `cyc` cycles
    printf("%-52s W=%d  %7.3f ms  %6.2f cycles/instr/SIMD  (%6.2f per wave)  clock %6.0f MHz  [event-time/2.4GHz: %5.2f]\n", kNames[KIND], wps, ms,
           cyc / (instr_per_wave * wps), cyc / instr_per_wave, mhz, ms * 1e-3 * 2.4e9 / (instr_per_wave * wps));
}

template <int K>
static void run_all(double* out, uint64_t* clocks) {
    for (int wps : {1, 2, 4}) run<K>(out, clocks, wps);
    if constexpr (K + 1 < NKINDS) run_all<K + 1>(out, clocks);
}

int main() {
    double* out;
    uint64_t* clocks;
    hipMalloc(&out, 256 * 8 * 256 * 8);
    hipMalloc(&clocks, 256 * 8 * 4 * 2 * 8);
    // bring the chip to its clock first
    for (int i = 0; i < 30; ++i) issue_kernel<FMA3><<<1024, 256>>>(out, clocks, 20000, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    run_all<0>(out, clocks);
    return 0;
}
