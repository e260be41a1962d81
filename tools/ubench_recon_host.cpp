// Host-side cost of the native reconstruction loop (csrc/nested_spd_reconstruction_solve.hip) without any launch: the evaluator is a
// quadratic in C, so that what is timed is the manifold arithmetic between two evaluations.  No GPU needed.
//   g++ -O2 tools/ubench_recon_host.cpp -Iinclude -Lgabotorch_amd -lgabo_hip -Wl,-rpath,$PWD/gabotorch_amd -o /tmp/ubench_recon_host && /tmp/ubench_recon_host 20 2
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gabo_hip.h"

struct Ctx { int D, d; std::vector<double> v0, c0, k0; long calls; };

static int quad(void* p, int64_t P, const double* v, const double* c, const double* k, double* cost, double* gv, double* gc, double* gk) {
    Ctx& x = *static_cast<Ctx*>(p);
    const int m = x.D - x.d, nV = x.D * m, nC = m * m, nK = x.d * m;
    x.calls += 1;
    for (int64_t q = 0; q < P; ++q) {
        double s = 0.0;
        for (int i = 0; i < nV; ++i) { const double t = v[q * nV + i] - x.v0[i]; gv[q * nV + i] = t; s += 0.5 * t * t; }
        for (int i = 0; i < nC; ++i) { const double t = c[q * nC + i] - x.c0[i]; gc[q * nC + i] = t; s += 0.5 * t * t; }
        for (int i = 0; i < nK; ++i) { const double t = k[q * nK + i] - x.k0[i]; gk[q * nK + i] = t; s += 0.5 * t * t; }
        cost[q] = s;
    }
    return GABO_OK;
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 20, d = argc > 2 ? atoi(argv[2]) : 2, m = D - d;
    srand(1);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    Ctx x{D, d, std::vector<double>(D * m), std::vector<double>(m * m), std::vector<double>(d * m), 0};
    // targets: some orthonormal-ish V0 (columns of the identity), SPD C0, small K0
    for (int i = 0; i < D; ++i) for (int j = 0; j < m; ++j) x.v0[i * m + j] = (i == j + d) ? 1.0 : 0.0;
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) x.c0[i * m + j] = (i == j ? 2.0 : 0.0) + 0.05 * (rnd() + 0.5) * (i == j);
    for (auto& t : x.k0) t = 0.2 * rnd();
    std::vector<double> w(D * d, 0.0), v(D * m, 0.0), c(m * m, 0.0), u(d * m), raw(1, 0.3);
    for (int j = 0; j < d; ++j) w[j * d + j] = 1.0;                                  // W = first d columns of the identity
    // start: V = a rotation of V0 mixing in W's span, C = identity, unit random
    for (int i = 0; i < D; ++i) for (int j = 0; j < m; ++j) v[i * m + j] = x.v0[i * m + j];
    const double th = 0.3;
    v[0 * m + 0] = std::sin(th); v[d * m + 0] = std::cos(th);
    for (int i = 0; i < m; ++i) c[i * m + i] = 1.0;
    double nn = 0.0;
    for (auto& t : u) { t = rnd(); nn += t * t; }
    for (auto& t : u) t /= std::sqrt(nn);
    gabo_recon_solve_options o{20, 1, 0.3, 0.8, 1e-3, 1e-6, 0.05, 1e-10, 1000, 6, 1e-10, 1000, INFINITY, 100, argc > 3 ? atoi(argv[3]) : 2, argc > 4 ? atoi(argv[4]) : 0};
    gabo_recon_solve_log log;
    const size_t npar = (size_t)D * m + m * m + d * m;
    std::vector<double> staging(GABO_RECON_MAX_LOOKAHEAD * (2 * npar + 1 + m + m * m));
    const std::vector<double> v_start = v, c_start = c, u_start = u;
    double sec = 1e30;
    int rc = 0;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 30;
    for (int rep = 0; rep < reps; ++rep) {                  // best of 30 identical runs
        v = v_start; c = c_start; u = u_start; raw[0] = 0.3;
        const auto t0 = std::chrono::steady_clock::now();
        rc = gabo_nested_spd_reconstruction_solve_with(quad, &x, w.data(), v.data(), c.data(), u.data(), raw.data(), D, d, staging.data(),
                                                       staging.size(), &o, &log);
        sec = std::min(sec, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("D=%d d=%d rc=%d: outer %ld, inner %ld, launches %ld, evaluations %ld, final cost %.3e, violation %.2e: %.2f ms = %.1f us per launch, %.1f us per inner iteration\n",
           D, d, rc, (long)log.outer_iterations, (long)log.inner_iterations, (long)log.launches, (long)log.evaluations, log.final_cost, log.violation,
           1e3 * sec, 1e6 * sec / log.launches, 1e6 * sec / (log.inner_iterations ? log.inner_iterations : 1));
    return 0;
}
