// What one dqds step costs next to one root-free QL step on gfx950, measured the way tools/ubench_issue.hip measures:
// in-kernel shader clocks, W waves per SIMD, the whole chip busy.  Both loops run on register arrays of order 10 with the reciprocal the
// product kernel uses (hardware seed + one Newton step, gabo_device.hpp: rcp_nr1); `sigma` changes per sweep so that nothing is hoisted.
//   dqds step  : q'_i = d + e_i;  t = q_{i+1} / q'_i;  e'_i = e_i t;  d = d t - sigma                (add, rcp + 2 fma, mul, mul, fma)
//   QL step    : the 15 VALU + reciprocal of csrc/spd_eig.hpp (form 2: gamma' = f (t p), p' = f^2 t + 1e-150)
// One implicit QL sweep is algebraically TWO Cholesky-LR (dqds) sweeps with the same shift, so the step ratio measured here has to be read
// against the step COUNTS of tools/sim/dqds_lapack_count.py / dqds_sim.py (profiles/r05_dqds_sim.txt).
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gabotorch_amd/csrc tools/ubench_dqds.hip -o /tmp/ubench_dqds && /tmp/ubench_dqds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "gabo_device.hpp"

using namespace gabo;
constexpr int D = 10;

template <int KIND>   // 0: dqds sweeps, 1: QL sweeps
__global__ __launch_bounds__(256, 2) void sweep_kernel(double* out, uint64_t* clocks, int sweeps, double seed) {
    double q[D], e[D];
    static_for<D>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        q[i] = 1.0 + 0.1 * i + seed * (threadIdx.x & 63);
        e[i] = 0.01 + 0.001 * i;
    });
    double sigma = 1e-3 * seed;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t r0 = wall_clock64();
    for (int s = 0; s < sweeps; ++s) {
        if constexpr (KIND == 0) {
            double d = q[0] - sigma;
            static_for<D - 1>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const double qq = d + e[i];
                const double t = q[i + 1] * rcp_nr1(qq);
                q[i] = qq;
                e[i] = e[i] * t;
                d = __builtin_fma(d, t, -sigma);
            });
            q[D - 1] = d;
            sigma = sigma * 0.999;                     // (a fresh shift per sweep)
            // keep the arrays in a benign range: undo the drift of repeated transforms without touching the instruction mix measured
            if ((s & 63) == 63) static_for<D>([&](auto ii) { q[decltype(ii)::value] = 1.0 + 0.1 * decltype(ii)::value; e[decltype(ii)::value] = 0.01; });
        } else {
            double gamma = q[D - 1] - sigma, p, sv = 0.0;
            asm("v_fma_f64 %0, %1, %1, %2" : "=v"(p) : "v"(gamma), "s"(1e-150));
            static_for_down<D - 2, 0>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const double bb = e[i];
                const double r = p + bb;
                if constexpr (i != D - 2) e[i + 1] = sv * r;
                const double t = rcp_nr1(p * r);
                const double ir = t * p;
                sv = bb * ir;
                const double oldgam = gamma, al = q[i];
                const double f = __builtin_fma(p, al - sigma, -(bb * oldgam));
                gamma = ir * f;
                q[i + 1] = oldgam + (al - gamma);
                const double ft = f * t;
                asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(ft), "v"(f), "s"(1e-150));
            });
            e[0] = sv * p;
            q[0] = sigma + gamma;
            sigma = sigma * 0.999;
            if ((s & 63) == 63) static_for<D>([&](auto ii) { q[decltype(ii)::value] = 1.0 + 0.1 * decltype(ii)::value; e[decltype(ii)::value] = 0.01; });
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t r1 = wall_clock64();
    double acc = 0;
    static_for<D>([&](auto ii) { acc += q[decltype(ii)::value] + e[decltype(ii)::value]; });
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = acc;
    if ((threadIdx.x & 63) == 0) {
        clocks[2 * (gid >> 6)] = t1 - t0;
        clocks[2 * (gid >> 6) + 1] = r1 - r0;
    }
}

template <int KIND>
static double run(double* out, uint64_t* clocks, int wps, const char* name) {
    const int blocks = 256 * wps, waves = blocks * 4, sweeps = 4000;
    sweep_kernel<KIND><<<blocks, 256>>>(out, clocks, 100, 1e-6);
    hipDeviceSynchronize();
    sweep_kernel<KIND><<<blocks, 256>>>(out, clocks, sweeps, 1e-6);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(2 * waves);
    hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; real += (double)h[2 * w + 1]; }
    cyc /= waves;
    real /= waves;
    const double steps = (double)sweeps * (D - 1);
    const double per_simd = cyc / (steps * wps);
    printf("%-28s W=%d  %7.1f cycles per step per wave, %6.1f per step per SIMD = %5.2f fp64 issue slots of 4.5 cycles; clock %5.0f MHz\n", name, wps, cyc / steps,
           per_simd, per_simd / 4.5, cyc / (real / 100.0));
    return per_simd;
}

int main() {
    double* out;
    uint64_t* clocks;
    hipMalloc(&out, 256 * 4 * 256 * 8);
    hipMalloc(&clocks, 256 * 4 * 4 * 2 * 8);
    for (int i = 0; i < 20; ++i) sweep_kernel<1><<<1024, 256>>>(out, clocks, 2000, 1e-6);
    hipDeviceSynchronize();
    for (int wps : {1, 2}) {
        const double a = run<0>(out, clocks, wps, "dqds step (order 10)");
        const double b = run<1>(out, clocks, wps, "root-free QL step (order 10)");
        printf("    ratio QL / dqds step: %.2f  (one QL sweep = two dqds sweeps with the same shift: break-even at 2.00)\n", b / a);
    }
    return 0;
}
