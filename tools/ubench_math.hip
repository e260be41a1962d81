// accuracy of acos_fast / exp_neg (gabo_device.hpp) against the host libm, on a dense sweep
#include "../gabotorch_amd/csrc/gabo_device.hpp"
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void kl(const double* x, double* l, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const gabo::LogRegs lr = gabo::LogRegs::load();
    if (i < n) l[i] = gabo::log_pos(x[i], lr);
}
__global__ void k(const double* x, double* a, double* e, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const gabo::MathRegs mt = gabo::MathRegs::load();
    if (i < n) { a[i] = gabo::acos_fast(x[i], mt); e[i] = gabo::exp_neg(-700.0 * (x[i] + 1.0) * 0.5, mt); }
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), ha(n), he(n);
    for (int i = 0; i < n; ++i) {
        double t = (i + 0.5) / n;
        h[i] = (i % 3 == 0) ? 1.0 - std::exp(-35.0 * t) : ((i % 3 == 1) ? -1.0 + std::exp(-35.0 * t) : 2.0 * t - 1.0);
        if (h[i] > 1 - 1e-15) h[i] = 1 - 1e-15;
        if (h[i] < -1 + 1e-15) h[i] = -1 + 1e-15;
    }
    double *x, *a, *e;
    hipMalloc(&x, n * 8); hipMalloc(&a, n * 8); hipMalloc(&e, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, a, e, n);
    hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(he.data(), e, n * 8, hipMemcpyDeviceToHost);
    double ea = 0, ee = 0;
    for (int i = 0; i < n; ++i) {
        double ra = std::acos(h[i]);
        ea = std::fmax(ea, std::fabs(ha[i] - ra) / ra);
        double arg = -700.0 * (h[i] + 1.0) * 0.5, re = std::exp(arg);
        if (re > 1e-300) ee = std::fmax(ee, std::fabs(he[i] - re) / re);
    }
    {   // log_pos on [1e-8, 1e8] and very close to 1
        std::vector<double> hx(n), hl(n);
        for (int i = 0; i < n; ++i) { double t = (i + 0.5) / n; hx[i] = (i % 2) ? std::exp(36.8 * t - 18.4) : 1.0 + (t - 0.5) * 1e-3; }
        double* l; hipMalloc(&l, n * 8);
        hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
        kl<<<n / 256, 256>>>(x, l, n);
        hipMemcpy(hl.data(), l, n * 8, hipMemcpyDeviceToHost);
        double el = 0;
        for (int i = 0; i < n; ++i) { double r = std::log(hx[i]); el = std::fmax(el, std::fabs(hl[i] - r) / std::fmax(std::fabs(r), 1e-300)); }
        printf("log_pos max rel err %.3e\n", el);
    }
    printf("acos_fast max rel err %.3e   exp_neg max rel err %.3e\n", ea, ee);
    return 0;
}
