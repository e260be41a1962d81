// Sustained fp64 VALU rate on gfx950: v_fma_f64 with C independent chains per lane, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ __launch_bounds__(256) void fma_rate(double* out, double a, double b, int iters) {
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
template <int CHAINS>
void run(double* out, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * waves_per_simd;    // 256 CUs x (256-thread block = one wave per SIMD)
    int iters = 40000 / CHAINS * 4;
    fma_rate<CHAINS><<<blocks, 256>>>(out, 1.0000001, 1e-9, 100);
    hipEventRecord(e0);
    fma_rate<CHAINS><<<blocks, 256>>>(out, 1.0000001, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)CHAINS * iters * 4.0 * blocks;     // wave-instructions
    double per_simd = insts / 1024.0;
    printf("chains %2d waves/SIMD %d: %.3f ms  %.1f TFLOP/s   %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", CHAINS,
           waves_per_simd, ms, 2.0 * 64 * insts / ms * 1e-9, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    double* out; hipMalloc(&out, 256 * 4096 * 8);
    for (int w : {1, 2, 4, 8}) { run<4>(out, w); run<8>(out, w); run<16>(out, w); }
    return 0;
}
