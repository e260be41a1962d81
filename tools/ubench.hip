// Micro-measurements on gfx950: accuracy of the v_rcp_f64 / v_rsq_f64 seeds, and sustained fp64 FMA rate.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void seeds(const double* x, double* r, double* q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); q[i] = __builtin_amdgcn_rsq(x[i]); }
}
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
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = std::exp((i / double(n)) * 40.0 - 20.0) * (1.0 + 1e-3 * (i % 977));
    double *x, *r, *q;
    hipMalloc(&x, n * 8); hipMalloc(&r, n * 8); hipMalloc(&q, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    seeds<<<n / 256, 256>>>(x, r, q, n);
    std::vector<double> hr(n), hq(n);
    hipMemcpy(hr.data(), r, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hq.data(), q, n * 8, hipMemcpyDeviceToHost);
    double er = 0, eq = 0;
    for (int i = 0; i < n; ++i) {
        er = std::fmax(er, std::fabs(hr[i] * h[i] - 1.0));
        eq = std::fmax(eq, std::fabs(hq[i] * std::sqrt(h[i]) - 1.0));
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f)   v_rsq_f64 max rel err %.3e (2^%.1f)\n", er, std::log2(er), eq, std::log2(eq));
    double* out; hipMalloc(&out, 256 * 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {
        int blocks = 256 * waves * 2;  // 256 CUs, 256-thread blocks = 1 wave per SIMD each
        int iters = 20000;
        fma_rate<8><<<blocks, 256>>>(out, 1.0000001, 1e-9, 100);
        hipEventRecord(e0);
        fma_rate<8><<<blocks, 256>>>(out, 1.0000001, 1e-9, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 2.0 * 8 * iters * 256.0 * blocks;
        printf("fp64 FMA: %d blocks x 256 thr, 8 chains: %.3f ms -> %.1f TFLOP/s\n", blocks, ms, flops / ms * 1e-9);
    }
    return 0;
}
