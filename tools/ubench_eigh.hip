// Shader-cycle cost of the two wave-per-matrix symmetric eigen-solvers on one d x d matrix in LDS (one block, one wave):
//   lds_jacobi (parallel-ordering Jacobi, lds_linalg.hpp) against wave_eigh (Householder + QL in registers, wave_eigh.hpp),
//   with the residual |A V - V diag(lam)|_max and |V^T V - I|_max checked on the host.
// Build & run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DGABO_EIGH_CLOCKS -I gabotorch_amd/csrc tools/ubench_eigh.hip -o /tmp/ubench_eigh && /tmp/ubench_eigh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "lds_linalg.hpp"

template <bool QL>
__global__ __launch_bounds__(64) void eig_kernel(const double* __restrict__ a, double* __restrict__ lam, double* __restrict__ v,
                                                 long long* __restrict__ cycles, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* A = lds;
    double* V = A + d * d;
    double* cs = V + d * d;
    gabo::lds_load(a + (size_t)blockIdx.x * d * d, A, d);
    const long long t0 = __builtin_readcyclecounter();
    gabo::lds_eigh<QL>(A, V, cs, d);
    const long long t1 = __builtin_readcyclecounter();
    for (int k = threadIdx.x; k < d; k += 64) lam[blockIdx.x * d + k] = A[k * d + k];
    for (int k = threadIdx.x; k < d * d; k += 64) v[(size_t)blockIdx.x * d * d + k] = V[k];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// extreme eigenpairs only (wave_eig_extremes): lam[0] = max, lam[1] = min, v rows 0 / 1 the vectors
__global__ __launch_bounds__(64) void ext_kernel(const double* __restrict__ a, double* __restrict__ lam, double* __restrict__ v,
                                                 long long* __restrict__ cycles, int d) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* A = lds;
    double* V = A + d * d;
    double* cs = V + d * d;
    gabo::lds_load(a + (size_t)blockIdx.x * d * d, A, d);
    const long long t0 = __builtin_readcyclecounter();
    gabo::wave_eig_extremes_any(A, V, cs, d);
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x < 2) lam[blockIdx.x * 2 + threadIdx.x] = cs[threadIdx.x];
    for (int k = threadIdx.x; k < 2 * d; k += 64) v[(size_t)blockIdx.x * 2 * d + k] = A[k];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    const int nmat = 8;
    printf("%4s %14s %14s %12s %12s\n", "d", "jacobi cycles", "QL cycles", "QL resid", "QL orth");
    for (int d = 2; d <= 32; ++d) {
        std::vector<double> a((size_t)nmat * d * d);
        for (int m = 0; m < nmat; ++m) {
            std::vector<double> g(d * d);
            for (auto& x : g) x = nd(rng);
            for (int r = 0; r < d; ++r)
                for (int c = 0; c < d; ++c) {
                    double s = r == c ? 0.1 : 0.0;
                    for (int k = 0; k < d; ++k) s += g[r * d + k] * g[c * d + k] / d;
                    a[(size_t)m * d * d + r * d + c] = s;
                }
        }
        double *da, *dl, *dv;
        long long* dc;
        hipMalloc(&da, a.size() * 8);
        hipMalloc(&dl, nmat * d * 8);
        hipMalloc(&dv, a.size() * 8);
        hipMalloc(&dc, nmat * 8);
        hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice);
        const size_t lds = (size_t)(2 * d * d + gabo::kJacobiScratch) * 8;
        double cyc[2] = {0, 0}, resid = 0, orth = 0;
        for (int ql = 0; ql < 2; ++ql) {
            if (ql && d < gabo::kWaveEighMinDim) continue;
            for (int rep = 0; rep < 3; ++rep) {
                if (ql) hipLaunchKernelGGL(eig_kernel<true>, dim3(nmat), dim3(64), lds, 0, da, dl, dv, dc, d);
                else hipLaunchKernelGGL(eig_kernel<false>, dim3(nmat), dim3(64), lds, 0, da, dl, dv, dc, d);
            }
            hipDeviceSynchronize();
            std::vector<long long> c(nmat);
            std::vector<double> lam(nmat * d), v(a.size());
            hipMemcpy(c.data(), dc, nmat * 8, hipMemcpyDeviceToHost);
            hipMemcpy(lam.data(), dl, nmat * d * 8, hipMemcpyDeviceToHost);
            hipMemcpy(v.data(), dv, a.size() * 8, hipMemcpyDeviceToHost);
            for (auto x : c) cyc[ql] += (double)x / nmat;
            if (ql) {
                for (int m = 0; m < nmat; ++m)
                    for (int r = 0; r < d; ++r)
                        for (int k = 0; k < d; ++k) {
                            double av = 0, vv = 0;
                            for (int c2 = 0; c2 < d; ++c2) {
                                av += a[(size_t)m * d * d + r * d + c2] * v[(size_t)m * d * d + c2 * d + k];
                                vv += v[(size_t)m * d * d + c2 * d + r] * v[(size_t)m * d * d + c2 * d + k];
                            }
                            resid = fmax(resid, fabs(av - v[(size_t)m * d * d + r * d + k] * lam[m * d + k]));
                            orth = fmax(orth, fabs(vv - (r == k ? 1.0 : 0.0)));
                        }
            }
        }
        printf("%4d %14.0f %14.0f %12.1e %12.1e", d, cyc[0], cyc[1], resid, orth);
        if (d >= gabo::kWaveEighMinDim) {
            // extremes against the full solve
            std::vector<double> lamq(nmat * d), lam2(nmat * 2), v2((size_t)nmat * 2 * d);
            std::vector<long long> c2(nmat);
            hipMemcpy(lamq.data(), dl, nmat * d * 8, hipMemcpyDeviceToHost);       // (the QL eigenvalues of the last launch above)
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(ext_kernel, dim3(nmat), dim3(64), lds, 0, da, dl, dv, dc, d);
            hipDeviceSynchronize();
            hipMemcpy(lam2.data(), dl, nmat * 2 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(v2.data(), dv, (size_t)nmat * 2 * d * 8, hipMemcpyDeviceToHost);
            hipMemcpy(c2.data(), dc, nmat * 8, hipMemcpyDeviceToHost);
            double ce = 0, dlam = 0, res2 = 0, nrm2 = 0;
            for (int m = 0; m < nmat; ++m) {
                ce += (double)c2[m] / nmat;
                double mx = lamq[m * d], mn = lamq[m * d];
                for (int k = 1; k < d; ++k) { mx = fmax(mx, lamq[m * d + k]); mn = fmin(mn, lamq[m * d + k]); }
                dlam = fmax(dlam, fmax(fabs(lam2[2 * m] - mx) / mx, fabs(lam2[2 * m + 1] - mn) / mx));
                for (int h = 0; h < 2; ++h) {
                    double nn = 0;
                    for (int r = 0; r < d; ++r) {
                        double av = 0;
                        for (int c2i = 0; c2i < d; ++c2i) av += a[(size_t)m * d * d + r * d + c2i] * v2[(size_t)m * 2 * d + h * d + c2i];
                        res2 = fmax(res2, fabs(av - lam2[2 * m + h] * v2[(size_t)m * 2 * d + h * d + r]));
                        nn += v2[(size_t)m * 2 * d + h * d + r] * v2[(size_t)m * 2 * d + h * d + r];
                    }
                    nrm2 = fmax(nrm2, fabs(nn - 1.0));
                }
            }
            printf("   extremes: %8.0f cycles, |dlam|/lmax %.1e, resid %.1e, |v|^2-1 %.1e", ce, dlam, res2, nrm2);
        }
#ifdef GABO_EIGH_CLOCKS
        long long ph[8];
        hipMemcpyFromSymbol(ph, HIP_SYMBOL(gabo_eigh_clk), sizeof(ph));
        if (d >= gabo::kWaveEighMinDim)
            printf("   block 0: tridiagonalise %lld, accumulate Q %lld, QL %lld | lane-group solver: multisection %lld, RQI %lld, back-transform %lld, rest %lld, [passes, solves, window] = %lld",
                   ph[1] - ph[0], ph[2] - ph[1], ph[3] - ph[2], ph[4] - ph[1], ph[5] - ph[4], ph[6] - ph[5], ph[2] - ph[6], ph[7]);
#endif
        printf("\n");
        hipFree(da); hipFree(dl); hipFree(dv); hipFree(dc);
    }
    return 0;
}
