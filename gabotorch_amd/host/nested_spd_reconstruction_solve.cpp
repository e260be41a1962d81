// Native host driver of the reconstruction-parameter optimisation of the nested-SPD mapping (HD-GaBO, config 5):
//   optimize_reconstruction_parameters_nested_spd   nested_mappings/nested_spd_optimization.py:95-186 (the run from the chosen start, :168-186)
//   AugmentedLagrangeMethod.solve                    manifold_optimization/augmented_Lagrange_method.py:72-326
//   [3P] pymanopt 0.2.x ConjugateGradient / LineSearchAdaptive (the reference's inner solver; SURVEY App. B: unpinned, restated)
// on the product  V in G(D, m) x C in S^m_++ x unit vector in S^(d m - 1) x raw in R,  K = sigmoid(raw) * unit reshaped d x m,  m = D - d,
// under ||V^T W||_F = 0.
//
// Why native: one evaluation of the cost + gradient is ONE launch of ~0.14 ms (gabo_nested_spd_reconstruction) and an optimisation asks
// for ~550 of them, one after the other (each needs the previous step's result): the numpy statement of the manifold operations around
// a launch (retractions with an 18 x 18 eigh and a polar factor, ~10 inner products with tr(C^-1 U C^-1 V), the bookkeeping of the
// look-ahead line search) cost as much wall-clock as the launches themselves.  Here they are a few thousand flops of straight C++ per
// conjugate-gradient iteration, and the evaluator is a function pointer: the built-in one is the HIP launch between two pinned staging
// copies; gabo_nested_spd_reconstruction_solve_with takes any other (a user-supplied cost_function, or the CPU tests' numpy cost).
// The algorithm is gabotorch_amd/manifold_optimization/{augmented_lagrange_method,conjugate_gradient,host_manifolds}.py statement by
// statement (same iterates up to the rounding of the eigen-solvers), including the look-ahead of the line search: the first trial step
// and its first contraction are evaluated by one launch.
// Host C++ only (no device code): compiled by the host compiler against the HIP runtime API (gabotorch_amd/_build.py, host/*.cpp).
#include <hip/hip_runtime_api.h>
#if defined(__linux__)
#include <sched.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/gabo_hip.h"

namespace gabo {

int nested_spd_reconstruction_launch(const double* data, const double* y, const double* sqrt_y, const double* w, const double* v,
                                     const double* c, const double* k, double* cost, double* grad_v, double* grad_c, double* grad_k,
                                     const double* c_eigenvalues, const double* c_eigenvectors, int64_t P, int64_t N, int D, int d, int metric,
                                     void* workspace, size_t workspace_bytes, bool clear_tickets, gabo_stream_t stream);     // csrc/nested_spd_reconstruction.hip

namespace host {

typedef std::vector<double> vec;

// The dense helpers below are compiled twice - baseline x86-64 and AVX2 - and the loader picks by the CPU it runs on (GNU ifunc): the
// loops are m^3-sized with m = D - d <= 31, the compiler's vectoriser is all they need.
#if defined(__x86_64__) && defined(__gnu_linux__) && !defined(GABO_HOST_NO_CLONES)
#define GABO_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define GABO_HOST_CLONES                                 // (no ifunc dispatch outside x86-64 glibc: one baseline build)
#endif

// (inlined into each clone of its callers: a call through the ifunc table per 18-element product would cost more than the product)
__attribute__((always_inline)) static inline double dot_inl(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

GABO_HOST_CLONES
static double dot(const double* a, const double* b, int n) { return dot_inl(a, b, n); }

// C (M x N) = A (M x K) B (K x N), row-major
GABO_HOST_CLONES
static void mm(const double* A, const double* B, double* C, int M, int K, int N) {
    for (int i = 0; i < M; ++i) {
        double* c = C + (size_t)i * N;
        for (int j = 0; j < N; ++j) c[j] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double a = A[(size_t)i * K + k];
            const double* b = B + (size_t)k * N;
            for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
    }
}

// C (M x N) = A^T B, A: K x M, B: K x N
GABO_HOST_CLONES
static void mtm(const double* A, const double* B, double* C, int K, int M, int N) {
    for (int i = 0; i < M * N; ++i) C[i] = 0.0;
    for (int k = 0; k < K; ++k) {
        const double* a = A + (size_t)k * M;
        const double* b = B + (size_t)k * N;
        for (int i = 0; i < M; ++i) {
            const double ai = a[i];
            double* c = C + (size_t)i * N;
            for (int j = 0; j < N; ++j) c[j] += ai * b[j];
        }
    }
}

// C (M x N) = A B^T, A: M x K, B: N x K
GABO_HOST_CLONES
static void mmt(const double* A, const double* B, double* C, int M, int K, int N) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) C[(size_t)i * N + j] = dot_inl(A + (size_t)i * K, B + (size_t)j * K, K);
}

static void symmetrize(double* A, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) A[i * n + j] = A[j * n + i] = 0.5 * (A[i * n + j] + A[j * n + i]);
}

// Eigen-decomposition of a symmetric n x n matrix: Householder reduction to tridiagonal form, then the implicit QL iteration with
// Wilkinson shifts carrying the transformation along.  a: the matrix (destroyed); w: eigenvalues (unsorted); q: eigenvectors as ROWS
// (q[i * n + k] = component k of the vector of w[i]) so that a rotation touches two contiguous rows.  e: 3 n doubles of scratch.
// false: no convergence.
GABO_HOST_CLONES
static bool sym_eig(int n, double* a, double* w, double* q, double* e) {
    // q = identity; reflectors are applied to its rows as they are found: q <- H q, so that in the end q A q^T is tridiagonal
    for (int i = 0; i < n * n; ++i) q[i] = 0.0;
    for (int i = 0; i < n; ++i) q[i * n + i] = 1.0;
    double* v = e + n;
    double* p = e + 2 * n;
    for (int k = 0; k + 2 < n; ++k) {
        const int len = n - k - 1;                       // the part of column k below the diagonal
        double scale = 0.0;
        for (int i = 0; i < len; ++i) scale = std::max(scale, std::fabs(a[(k + 1 + i) * n + k]));
        if (scale == 0.0) continue;
        double norm2 = 0.0;
        for (int i = 0; i < len; ++i) { v[i] = a[(k + 1 + i) * n + k] / scale; norm2 += v[i] * v[i]; }
        const double alpha = v[0] >= 0.0 ? -std::sqrt(norm2) : std::sqrt(norm2);
        const double vnorm2 = norm2 - v[0] * alpha;      // ||v - alpha e1||^2 / 2
        v[0] -= alpha;
        if (vnorm2 == 0.0) continue;
        const double inv = 1.0 / vnorm2;                 // H = I - v v^T / vnorm2
        // trailing block B (rows / columns k+1..n-1): B <- H B H
        for (int i = 0; i < len; ++i) p[i] = inv * dot_inl(a + (size_t)(k + 1 + i) * n + k + 1, v, len);
        const double kk = 0.5 * inv * dot_inl(p, v, len);
        for (int i = 0; i < len; ++i) p[i] -= kk * v[i];
        for (int i = 0; i < len; ++i) {
            double* row = a + (size_t)(k + 1 + i) * n + k + 1;
            const double vi = v[i], pi = p[i];
            for (int j = 0; j < len; ++j) row[j] -= vi * p[j] + pi * v[j];
        }
        a[(k + 1) * n + k] = a[k * n + k + 1] = alpha * scale;
        for (int i = 1; i < len; ++i) a[(k + 1 + i) * n + k] = a[k * n + k + 1 + i] = 0.0;
        // rows k+1.. of q: q <- H q  (row by row: contiguous, vectorisable; p is free again and holds v^T q / vnorm2)
        for (int j = 0; j < n; ++j) p[j] = 0.0;
        for (int i = 0; i < len; ++i) {
            const double vi = v[i];
            const double* row = q + (size_t)(k + 1 + i) * n;
            for (int j = 0; j < n; ++j) p[j] += vi * row[j];
        }
        for (int j = 0; j < n; ++j) p[j] *= inv;
        for (int i = 0; i < len; ++i) {
            const double vi = v[i];
            double* row = q + (size_t)(k + 1 + i) * n;
            for (int j = 0; j < n; ++j) row[j] -= vi * p[j];
        }
    }
    for (int i = 0; i < n; ++i) w[i] = a[i * n + i];
    for (int i = 0; i + 1 < n; ++i) e[i] = a[(i + 1) * n + i];
    e[n - 1] = 0.0;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int l = 0; l < n; ++l) {
        for (int iter = 0;; ++iter) {
            int m = l;
            for (; m + 1 < n; ++m)
                if (std::fabs(e[m]) <= eps * (std::fabs(w[m]) + std::fabs(w[m + 1]))) break;
            if (m == l) break;
            if (iter == 80) return false;
            double g = (w[l + 1] - w[l]) / (2.0 * e[l]);
            double r = std::sqrt(g * g + 1.0);
            g = w[m] - w[l] + e[l] / (g + (g >= 0.0 ? r : -r));
            double s = 1.0, c = 1.0, p2 = 0.0;
            int i = m - 1;
            for (; i >= l; --i) {
                const double f = s * e[i], b = c * e[i];
                r = std::sqrt(f * f + g * g);
                e[i + 1] = r;
                if (r == 0.0) { w[i + 1] -= p2; e[m] = 0.0; break; }
                s = f / r;
                c = g / r;
                g = w[i + 1] - p2;
                r = (w[i] - g) * s + 2.0 * c * b;
                p2 = s * r;
                w[i + 1] = g + p2;
                g = c * r - b;
                double* q0 = q + (size_t)i * n;
                double* q1 = q + (size_t)(i + 1) * n;
                for (int k = 0; k < n; ++k) {
                    const double t = q1[k];
                    q1[k] = s * q0[k] + c * t;
                    q0[k] = c * q0[k] - s * t;
                }
            }
            if (r == 0.0 && i >= l) continue;
            w[l] -= p2;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    return true;
}

// L lower-triangular with L L^T = A (row-major, upper part of L zeroed).  false: not positive definite.
static bool cholesky(const double* A, double* L, int n) {
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j] - dot(L + (size_t)j * n, L + (size_t)j * n, j);
        if (!(s > 0.0)) return false;
        const double ljj = std::sqrt(s);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) L[i * n + j] = (A[i * n + j] - dot(L + (size_t)i * n, L + (size_t)j * n, j)) / ljj;
    }
    return true;
}

// Li = L^-1 (lower-triangular)
static void lower_inverse(const double* L, double* Li, int n) {
    for (int i = 0; i < n * n; ++i) Li[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += L[i * n + k] * Li[k * n + j];
            Li[i * n + j] = -s / L[i * n + i];
        }
    }
}

enum { kMaxLookahead = GABO_RECON_MAX_LOOKAHEAD };

struct Dims {
    int D, d, m;
    int nV, nC, nU, n;          // sizes of the factors and of a point / tangent vector [V | C | unit | raw]
    int oC, oU, oS;
    Dims(int D_, int d_) : D(D_), d(d_), m(D_ - d_) {
        nV = D * m; nC = m * m; nU = d * m; n = nV + nC + nU + 1;
        oC = nV; oU = nV + nC; oS = nV + nC + nU;
    }
};

// A point of the product with what has been computed at it
struct Point {
    vec x;                      // [V (D x m) | C (m x m) | unit (d m) | raw]
    vec egrad;                  // Euclidean gradient of the reconstruction cost (chained through K = sigmoid(raw) unit)
    double f = 0.0;             // reconstruction cost
    double g = 0.0;             // constraint ||V^T W||_F
    vec cinv;                   // C^-1, made when the first inner product at this point asks for it
    bool has_cinv = false;
    vec c_lam, c_vec;           // eigenvalues / eigenvectors (in the columns) of sym(C), for evaluators that take them along (the HIP launch)
    bool has_factors = false;
    explicit Point(const Dims& dm) : x(dm.n), egrad(dm.n), cinv(dm.nC), c_lam(dm.m), c_vec(dm.nC) {}
};

// Scratch of one line-search candidate: the candidates of a launch are retracted (and their C factored) side by side on the pool's threads.
struct StepScratch {
    vec a, m0, m1, m2, w, e;
    int error = GABO_OK;
    explicit StepScratch(const Dims& dm) : a(dm.nV), m0(dm.nC), m1(dm.nC), m2(dm.nC), w(dm.m), e(3 * dm.m) {}
};

// A handful of worker threads that spin between the launches of ONE optimisation (tens of milliseconds): task 0 runs on the caller,
// tasks 1.. on the workers; every worker answers every run (no straggler can meet the next run's job).
class StepPool {
    std::vector<std::thread> workers;
    std::atomic<unsigned> generation{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};
    const std::function<void(int)>* job = nullptr;
    int ntasks = 0;

    // spin politely; on an oversubscribed machine (fewer runnable cores than threads) give the time slice away instead of burning it
    static void relax(unsigned& spins) {
#if defined(__x86_64__)
        if (++spins < (1u << 20)) __builtin_ia32_pause();      // (~tens of ms: longer than any wait inside one optimisation)
        else std::this_thread::yield();
#else
        ++spins;
        std::this_thread::yield();
#endif
    }

 public:
    explicit StepPool(int nworkers) {
        for (int w = 0; w < nworkers; ++w)
            workers.emplace_back([this, w] {
                unsigned seen = 0;
                while (true) {
                    unsigned spins = 0;
                    while (generation.load(std::memory_order_acquire) == seen) {
                        if (stop.load(std::memory_order_relaxed)) return;
                        relax(spins);
                    }
                    seen += 1;
                    if (w + 1 < ntasks) (*job)(w + 1);
                    pending.fetch_sub(1, std::memory_order_release);
                }
            });
    }
    int threads() const { return (int)workers.size() + 1; }
    void run(int n, const std::function<void(int)>& f) {
        if (workers.empty() || n <= 1) {
            for (int i = 0; i < n; ++i) f(i);
            return;
        }
        job = &f;
        ntasks = n;
        pending.store((int)workers.size(), std::memory_order_relaxed);
        generation.fetch_add(1, std::memory_order_release);
        f(0);
        for (int i = (int)workers.size() + 1; i < n; ++i) f(i);          // (more tasks than threads: the rest here)
        unsigned spins = 0;
        while (pending.load(std::memory_order_acquire) > 0) relax(spins);
    }
    ~StepPool() {
        stop.store(true);
        for (auto& t : workers) t.join();
    }
};

struct Driver {
    Dims dm;
    const double* W;                                   // D x d, host
    gabo_recon_eval_fn eval;
    void* ctx;
    double* stage_in;                                  // kMaxLookahead * (nV + nC + nU + factors): [V x P | C x P | K x P]
    double* stage_out;                                 // kMaxLookahead * (1 + nV + nC + nU): [cost x P | gV x P | gC x P | gK x P]
    gabo_recon_solve_options opt;
    double rho = 1.0, gamma = 1.0;
    int lookahead = 2;
    int64_t evaluations = 0, launches = 0, inner_iterations = 0;
    double seconds_evaluator = 0.0;
    int error = GABO_OK;
    vec t0, t1, t2, t3, t4, ew, ee;                    // scratch
    bool stage_factors = false;                        // the evaluator takes the eigen-decomposition of every C behind the parameter sets
    std::vector<StepScratch> steps_scratch;
    StepPool* pool = nullptr;

    Driver(int D, int d, const double* w, gabo_recon_eval_fn fn, void* c, double* in, double* out, const gabo_recon_solve_options& o)
        : dm(D, d), W(w), eval(fn), ctx(c), stage_in(in), stage_out(out), opt(o) {
        const int big = std::max(dm.nV, dm.nC) + dm.D * dm.D;
        t0.resize(big); t1.resize(big); t2.resize(big); t3.resize(big); t4.resize(big);
        ew.resize(dm.D); ee.resize(3 * dm.D);
        for (int i = 0; i < kMaxLookahead; ++i) steps_scratch.emplace_back(dm);
    }

    // eigen-decomposition of sym(C) of a point (vectors in the columns), with the scratch of one candidate slot
    bool factor_point(Point& pt, StepScratch& sc) {
        const int m = dm.m;
        const double* cp = pt.x.data() + dm.oC;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) sc.m0[i * m + j] = 0.5 * (cp[i * m + j] + cp[j * m + i]);
        if (!sym_eig(m, sc.m0.data(), pt.c_lam.data(), sc.m1.data(), sc.e.data())) return false;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) pt.c_vec[i * m + j] = sc.m1[j * m + i];
        pt.has_factors = true;
        return true;
    }

    // ---------------------------------------------------------------------------------------------- the evaluator
    // reconstruction cost + Euclidean gradient (and the constraint value) at P <= kMaxLookahead points, ONE call of the evaluator
    void evaluate(Point** pts, int P) {
        const int npar = dm.nV + dm.nC + dm.nU;
        double ts[kMaxLookahead];
        for (int p = 0; p < P; ++p) {
            const double* x = pts[p]->x.data();
            ts[p] = 1.0 / (1.0 + std::exp(-x[dm.oS]));                                  // gpytorch Interval(0, 1).transform (:139, 155)
            std::memcpy(stage_in + (size_t)p * dm.nV, x, sizeof(double) * dm.nV);
            std::memcpy(stage_in + (size_t)P * dm.nV + (size_t)p * dm.nC, x + dm.oC, sizeof(double) * dm.nC);
            double* k = stage_in + (size_t)P * (dm.nV + dm.nC) + (size_t)p * dm.nU;
            for (int i = 0; i < dm.nU; ++i) k[i] = ts[p] * x[dm.oU + i];
            if (stage_factors) {
                if (!pts[p]->has_factors && !factor_point(*pts[p], steps_scratch[0]) && error == GABO_OK) error = GABO_ERR_NOT_SPD;
                std::memcpy(stage_in + (size_t)P * npar + (size_t)p * dm.m, pts[p]->c_lam.data(), sizeof(double) * dm.m);
                std::memcpy(stage_in + (size_t)P * (npar + dm.m) + (size_t)p * dm.nC, pts[p]->c_vec.data(), sizeof(double) * dm.nC);
            }
        }
        double* cost = stage_out;
        double* gv = stage_out + P;
        double* gc = gv + (size_t)P * dm.nV;
        double* gk = gc + (size_t)P * dm.nC;
        const auto t_eval = std::chrono::steady_clock::now();
        const int rc = eval(ctx, P, stage_in, stage_in + (size_t)P * dm.nV, stage_in + (size_t)P * (dm.nV + dm.nC), cost, gv, gc, gk);
        seconds_evaluator += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_eval).count();
        if (rc != GABO_OK && error == GABO_OK) error = rc;
        launches += 1;
        evaluations += P;
        for (int p = 0; p < P; ++p) {
            Point& pt = *pts[p];
            const double* x = pt.x.data();
            pt.f = cost[p];
            std::memcpy(pt.egrad.data(), gv + (size_t)p * dm.nV, sizeof(double) * dm.nV);
            std::memcpy(pt.egrad.data() + dm.oC, gc + (size_t)p * dm.nC, sizeof(double) * dm.nC);
            const double* gkp = gk + (size_t)p * dm.nU;
            double s = 0.0;
            for (int i = 0; i < dm.nU; ++i) {
                pt.egrad[dm.oU + i] = ts[p] * gkp[i];
                s += gkp[i] * x[dm.oU + i];
            }
            pt.egrad[dm.oS] = s * ts[p] * (1.0 - ts[p]);
            pt.g = constraint(x);
            pt.has_cinv = false;
        }
    }

    // ||V^T W||_F (:142-147)
    double constraint(const double* x) {
        mtm(W, x, t0.data(), dm.D, dm.d, dm.m);                 // W^T V: d x m
        return std::sqrt(dot(t0.data(), t0.data(), dm.d * dm.m));
    }

    // ---------------------------------------------------------------------------------------------- manifold operations
    const double* cinv(Point& p) {
        if (!p.has_cinv) {
            const int m = dm.m;
            double* L = t3.data();
            double* Li = t4.data();
            if (!cholesky(p.x.data() + dm.oC, L, m)) {
                if (error == GABO_OK) error = GABO_ERR_NOT_SPD;
                for (int i = 0; i < m * m; ++i) p.cinv[i] = std::numeric_limits<double>::quiet_NaN();
            } else {
                lower_inverse(L, Li, m);
                mtm(Li, Li, p.cinv.data(), m, m, m);               // C^-1 = Li^T Li
            }
            p.has_cinv = true;
        }
        return p.cinv.data();
    }

    double inner(Point& p, const double* u, const double* v) {
        const int m = dm.m;
        double s = dot(u, v, dm.nV) + dot(u + dm.oU, v + dm.oU, dm.nU) + u[dm.oS] * v[dm.oS];
        const double* ci = cinv(p);
        mm(ci, u + dm.oC, t0.data(), m, m, m);
        if (u == v) {
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < m; ++j) s += t0[j * m + i] * t0[i * m + j];      // tr(C^-1 U C^-1 U)
        } else {
            mm(ci, v + dm.oC, t1.data(), m, m, m);
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < m; ++j) s += t0[j * m + i] * t1[i * m + j];
        }
        return s;
    }

    double norm(Point& p, const double* u) { return std::sqrt(std::max(inner(p, u, u), 0.0)); }

    // tangent-space projection of the Grassmann / sphere factors at x (identity on the other two): out may alias u
    void project_ambient(const double* x, const double* u, double* out) {
        const int D = dm.D, m = dm.m;
        mtm(x, u, t0.data(), D, m, m);                          // V^T U
        mm(x, t0.data(), t1.data(), D, m, m);
        for (int i = 0; i < dm.nV; ++i) out[i] = u[i] - t1[i];
        const double xu = dot(x + dm.oU, u + dm.oU, dm.nU);
        for (int i = 0; i < dm.nU; ++i) out[dm.oU + i] = u[dm.oU + i] - xu * x[dm.oU + i];
    }

    void egrad2rgrad(const double* x, const double* g, double* out) {
        const int m = dm.m;
        project_ambient(x, g, out);
        const double* C = x + dm.oC;
        double* S = t0.data();
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) S[i * m + j] = 0.5 * (g[dm.oC + i * m + j] + g[dm.oC + j * m + i]);
        mm(C, S, t1.data(), m, m, m);
        mm(t1.data(), C, out + dm.oC, m, m, m);                 // C sym(G) C
        out[dm.oS] = g[dm.oS];
    }

    // transport to x2: projection on the Grassmann / sphere factors, identity on the others
    void transp(const double* x2, const double* u, double* out) {
        project_ambient(x2, u, out);
        if (out != u) {
            std::memcpy(out + dm.oC, u + dm.oC, sizeof(double) * dm.nC);
            out[dm.oS] = u[dm.oS];
        }
    }

    // outs[i] = retraction of steps[i] * u at x: polar factor of V + t U; L expm(t L^-1 U L^-T) L^T (one Cholesky factor and one
    // eigen-decomposition for every step length); (unit + t u) normalised; raw + t u
    void retr_steps(const double* x, const double* u, const double* steps, int nsteps, Point** outs) {
        const int D = dm.D, m = dm.m;
        // SPD factor
        double* L = t0.data();
        double* Li = t1.data();
        double* S = t2.data();
        double* Q = t3.data();
        double* LQ = t4.data();
        bool ok = cholesky(x + dm.oC, L, m);
        if (ok) {
            lower_inverse(L, Li, m);
            mm(Li, u + dm.oC, S, m, m, m);
            mmt(S, Li, Q, m, m, m);                             // Li U Li^T
            symmetrize(Q, m);
            std::memcpy(S, Q, sizeof(double) * m * m);
            ok = sym_eig(m, S, ew.data(), Q, ee.data());        // rows of Q: eigenvectors
            mmt(L, Q, LQ, m, m, m);                             // LQ[i][k] = sum_j L[i][j] Q[k][j] = (L Qcols)[i][k]
        }
        if (!ok && error == GABO_OK) error = GABO_ERR_NOT_SPD;
        // every step length: its SPD point, its Grassmann point (polar factor: A (A^T A)^-1/2), sphere / Euclidean factors and - for the
        // evaluators that take it - the eigen-decomposition of the new C; independent of each other: one task each on the pool's threads
        const double* LQc = LQ;
        const double* ews = ew.data();
        const std::function<void(int)> task = [&, LQc, ews, ok](int s) {
            StepScratch& sc = steps_scratch[s];
            Point& out = *outs[s];
            double* y = out.x.data();
            double* C2 = y + dm.oC;
            if (!ok) {
                for (int i = 0; i < m * m; ++i) C2[i] = std::numeric_limits<double>::quiet_NaN();
            } else {
                double* scaled = sc.m0.data();                   // LQ diag(exp(t w))
                for (int i = 0; i < m; ++i)
                    for (int k = 0; k < m; ++k) scaled[i * m + k] = LQc[i * m + k] * std::exp(steps[s] * ews[k]);
                mmt(scaled, LQc, C2, m, m, m);
                symmetrize(C2, m);
            }
            double* A = sc.a.data();
            for (int i = 0; i < dm.nV; ++i) A[i] = x[i] + steps[s] * u[i];
            double* M = sc.m0.data();
            mtm(A, A, M, D, m, m);
            symmetrize(M, m);
            double* Qm = sc.m1.data();
            if (!sym_eig(m, M, sc.w.data(), Qm, sc.e.data())) sc.error = GABO_ERR_NOT_SPD;
            double* R = sc.m2.data();                            // (A^T A)^-1/2 = sum_k w_k^-1/2 q_k q_k^T
            for (int i = 0; i < m * m; ++i) R[i] = 0.0;
            for (int k = 0; k < m; ++k) {
                const double f = 1.0 / std::sqrt(sc.w[k]);
                const double* qk = Qm + (size_t)k * m;
                for (int i = 0; i < m; ++i) {
                    const double fi = f * qk[i];
                    for (int j = 0; j < m; ++j) R[i * m + j] += fi * qk[j];
                }
            }
            mm(A, R, y, D, m, m);
            double nn = 0.0;
            for (int i = 0; i < dm.nU; ++i) { y[dm.oU + i] = x[dm.oU + i] + steps[s] * u[dm.oU + i]; nn += y[dm.oU + i] * y[dm.oU + i]; }
            nn = std::sqrt(nn);
            for (int i = 0; i < dm.nU; ++i) y[dm.oU + i] /= nn;
            y[dm.oS] = x[dm.oS] + steps[s] * u[dm.oS];
            out.has_cinv = false;
            out.has_factors = false;
            if (stage_factors && ok && !factor_point(out, sc)) sc.error = GABO_ERR_NOT_SPD;
        };
        if (pool) pool->run(nsteps, task);
        else for (int s = 0; s < nsteps; ++s) task(s);
        for (int s = 0; s < nsteps; ++s)
            if (steps_scratch[s].error != GABO_OK && error == GABO_OK) error = steps_scratch[s].error;
    }

    // geodesic distance on the product (the augmented Lagrangian's step length between outer iterations, :199)
    double dist(const double* x, const double* y) {
        const int D = dm.D, m = dm.m;
        double total = 0.0;
        // Grassmann: principal angles = acos of the singular values of x^T y
        double* M = t0.data();
        mtm(x, y, M, D, m, m);
        double* MtM = t1.data();
        mtm(M, M, MtM, m, m, m);
        symmetrize(MtM, m);
        if (sym_eig(m, MtM, ew.data(), t2.data(), ee.data()))
            for (int k = 0; k < m; ++k) {
                const double c = std::min(1.0, std::sqrt(std::max(ew[k], 0.0)));
                const double a = std::acos(c);
                total += a * a;
            }
        // SPD: sqrt(sum log^2 lambda(L^-1 Y L^-T))
        double* L = t0.data();
        double* Li = t1.data();
        if (cholesky(x + dm.oC, L, m)) {
            lower_inverse(L, Li, m);
            mm(Li, y + dm.oC, t2.data(), m, m, m);
            mmt(t2.data(), Li, t3.data(), m, m, m);
            symmetrize(t3.data(), m);
            if (sym_eig(m, t3.data(), ew.data(), t2.data(), ee.data()))
                for (int k = 0; k < m; ++k) { const double l = std::log(ew[k]); total += l * l; }
        }
        const double c = std::max(-1.0, std::min(1.0, dot(x + dm.oU, y + dm.oU, dm.nU)));
        const double a = std::acos(c);
        total += a * a;
        const double ds = x[dm.oS] - y[dm.oS];
        total += ds * ds;
        return std::sqrt(total);
    }

    // ---------------------------------------------------------------------------------------------- the subproblem (:226-326)
    double sub_cost(const Point& p) const {
        const double t = gamma / rho + p.g;
        return p.f + 0.5 * rho * t * t;
    }

    // Riemannian gradient of the subproblem at p: grad f + (g rho + gamma) grad g,  egrad g = W (W^T V) / g on the V factor
    void sub_grad(Point& p, double* out) {
        const int D = dm.D, d = dm.d, m = dm.m;
        egrad2rgrad(p.x.data(), p.egrad.data(), out);
        if (p.g > 0.0) {
            const double coef = (p.g * rho + gamma) / p.g;
            double* wtv = t2.data();
            double* gv = t3.data();
            mtm(W, p.x.data(), wtv, D, d, m);
            mm(W, wtv, gv, D, d, m);
            // projection onto the tangent space of the Grassmann factor
            mtm(p.x.data(), gv, t0.data(), D, m, m);
            mm(p.x.data(), t0.data(), t1.data(), D, m, m);
            for (int i = 0; i < dm.nV; ++i) out[i] += coef * (gv[i] - t1[i]);
        }
    }

    // ---------------------------------------------------------------------------------------------- conjugate gradients
    // [3P] pymanopt ConjugateGradient with LineSearchAdaptive (Hestenes-Stiefel beta clipped at 0; sufficient decrease 1/2, contraction
    // 1/2, at most 10 cost evaluations; next initial step = last accepted, doubled unless it took exactly one contraction).
    // cur: in = start (evaluated), out = result (evaluated).  Returns the number of iterations.
    int64_t cg(Point& cur, double tolgradnorm, std::vector<Point>& cands) {
        using clock = std::chrono::steady_clock;
        const auto time0 = clock::now();
        const int n = dm.n;
        vec grad(n), desc(n), newgrad(n), oldgrad(n), diff(n), tmp(n);
        double cost = sub_cost(cur);
        sub_grad(cur, grad.data());
        double gradnorm = norm(cur, grad.data());
        double grad_grad = inner(cur, grad.data(), grad.data());
        for (int i = 0; i < n; ++i) desc[i] = -grad[i];
        double stepsize = std::numeric_limits<double>::quiet_NaN();
        double oldalpha = -1.0;
        int64_t it = 0;
        while (true) {
            if (gradnorm < tolgradnorm) break;
            if (it >= opt.cg_maxiter) break;
            if (std::chrono::duration<double>(clock::now() - time0).count() >= opt.cg_maxtime) break;
            if (stepsize < opt.cg_minstepsize) break;
            if (error != GABO_OK) break;
            double df0 = inner(cur, grad.data(), desc.data());
            if (df0 >= 0.0) {                                   // not a descent direction: restart from steepest descent
                for (int i = 0; i < n; ++i) desc[i] = -grad[i];
                df0 = -grad_grad;
            }
            // ---- line search: the first trial step and its first `look - 1` contractions in one launch (the blocks of different
            // parameter sets run side by side: the launch takes as long as a single evaluation)
            const double norm_d = norm(cur, desc.data());
            double alpha = oldalpha >= 0.0 ? oldalpha : 1.0 / norm_d;
            const int look = (int)cands.size();
            double steps[kMaxLookahead];
            Point* ahead[kMaxLookahead];
            for (int i = 0; i < look; ++i) { steps[i] = i == 0 ? alpha : 0.5 * steps[i - 1]; ahead[i] = &cands[i]; }
            retr_steps(cur.x.data(), desc.data(), steps, look, ahead);
            evaluate(ahead, look);
            Point* cand = &cands[0];
            double newf = sub_cost(*cand);
            int evals = 1;
            while (newf > cost + 0.5 * alpha * df0 && evals <= 10) {
                alpha *= 0.5;
                if (evals < look) {
                    cand = &cands[evals];
                } else {
                    Point* one[1] = {&cands[0]};
                    retr_steps(cur.x.data(), desc.data(), &alpha, 1, one);
                    evaluate(one, 1);
                    cand = &cands[0];
                }
                newf = sub_cost(*cand);
                evals += 1;
            }
            const bool rejected = newf > cost;
            if (rejected) { alpha = 0.0; newf = cost; }
            oldalpha = evals == 2 ? alpha : 2.0 * alpha;
            stepsize = alpha * norm_d;
            Point& nx = rejected ? cur : *cand;
            // ---- direction update
            sub_grad(nx, newgrad.data());
            const double newgradnorm = norm(nx, newgrad.data());
            const double new_gg = inner(nx, newgrad.data(), newgrad.data());
            transp(nx.x.data(), grad.data(), oldgrad.data());
            const double orth = new_gg > 0.0 ? inner(nx, oldgrad.data(), newgrad.data()) / new_gg : 0.0;
            if (std::fabs(orth) >= opt.cg_orth_value) {
                for (int i = 0; i < n; ++i) desc[i] = -newgrad[i];
            } else {
                transp(nx.x.data(), desc.data(), tmp.data());
                for (int i = 0; i < n; ++i) diff[i] = newgrad[i] - oldgrad[i];
                const double den = inner(nx, diff.data(), tmp.data());
                const double beta = den != 0.0 ? std::max(0.0, inner(nx, newgrad.data(), diff.data()) / den) : 1.0;   // Hestenes-Stiefel
                for (int i = 0; i < n; ++i) desc[i] = beta * tmp[i] - newgrad[i];
            }
            if (!rejected) {
                std::swap(cur.x, cand->x);
                std::swap(cur.egrad, cand->egrad);
                std::swap(cur.cinv, cand->cinv);
                std::swap(cur.c_lam, cand->c_lam);
                std::swap(cur.c_vec, cand->c_vec);
                cur.f = cand->f; cur.g = cand->g; cur.has_cinv = cand->has_cinv; cur.has_factors = cand->has_factors;
            }
            cost = newf;
            grad.swap(newgrad);
            gradnorm = newgradnorm;
            grad_grad = new_gg;
            it += 1;
        }
        return it;
    }

    // ---------------------------------------------------------------------------------------------- augmented Lagrangian (:72-225)
    void solve(Point& best, gabo_recon_solve_log* log) {
        using clock = std::chrono::steady_clock;
        const auto time0 = clock::now();
        Point prev(dm);
        std::vector<Point> cands((size_t)lookahead, Point(dm));
        rho = opt.rho_init;
        gamma = opt.gammas_fact;
        double oldacc = std::numeric_limits<double>::infinity();
        double tol = opt.starting_tolgradnorm;
        const double theta_tol = std::pow(opt.ending_tolgradnorm / opt.starting_tolgradnorm, 1.0 / (double)opt.maxiter);
        int64_t k = 0;
        int reason = GABO_RECON_STOP_MAXITER;
        Point* start[1] = {&best};
        evaluate(start, 1);
        prev.x = best.x;
        while (error == GABO_OK) {
            inner_iterations += cg(best, tol, cands);
            const double v = best.g;                            // (:185-188) the one equality constraint
            const double newacc = std::fabs(v);
            gamma = std::min(opt.bound, std::max(-opt.bound, gamma + rho * v));
            if (k == 0 || newacc > opt.tau * oldacc) rho = rho / opt.thetarho;     // (:191-193)
            oldacc = newacc;
            tol = std::max(opt.ending_tolgradnorm, tol * theta_tol);
            k += 1;
            const double step = dist(best.x.data(), prev.x.data());
            if (std::chrono::duration<double>(clock::now() - time0).count() >= opt.maxtime) { reason = GABO_RECON_STOP_MAXTIME; break; }
            if (k >= opt.maxiter) { reason = GABO_RECON_STOP_MAXITER; break; }
            if (step < opt.minstepsize) { reason = GABO_RECON_STOP_MINSTEP; break; }
            if (tol <= opt.ending_tolgradnorm) { reason = GABO_RECON_STOP_MINGRAD; break; }
            prev.x = best.x;
        }
        if (log) {
            log->outer_iterations = k;
            log->inner_iterations = inner_iterations;
            log->evaluations = evaluations;
            log->launches = launches;
            log->stop_reason = reason;
            log->violation = oldacc;
            log->rho = rho;
            log->gamma = gamma;
            log->final_cost = best.f;
            log->seconds = std::chrono::duration<double>(clock::now() - time0).count();
            log->seconds_evaluator = seconds_evaluator;
        }
    }
};

static bool options_ok(const gabo_recon_solve_options* o) {
    return o && o->maxiter >= 1 && o->cg_maxiter >= 0 && o->rho_init > 0.0 && o->thetarho > 0.0 && o->starting_tolgradnorm > 0.0 &&
           o->ending_tolgradnorm > 0.0;
}

// The built-in evaluator: pinned staging -> device, one launch, device -> pinned staging, wait.  (Letting the kernel read and write the
// page-locked staging memory directly instead of the two small copies was measured: no difference, 45.3 against 45.8 ms per optimisation
// at D = 20 - the copies are not what the ~30 us between the end of the kernel and the host's next instruction are made of.)
// The eigen-decomposition of each C goes along with the parameters: ~5 us of host arithmetic here against a lone wave's ~50 us on
// the critical path of every block there.
struct HipEvaluator {
    const double *data, *y, *sqrt_y, *w;
    int64_t N;
    int D, d, metric;
    double *dev_in, *dev_out;
    void* recon_ws;
    size_t recon_ws_bytes;
    hipStream_t stream;
    bool factors_staged = false;          // the driver put the eigen-decomposition of every C behind the parameter sets already
    vec a, q, lam, e;
    double t_factor = 0.0, t_enqueue = 0.0, t_wait = 0.0;     // where an evaluation's wall-clock goes (reported with GABO_RECON_TIMING set)
};

static int hip_evaluate(void* ctx, int64_t P, const double* v, const double* c, const double* k, double* cost, double* gv, double* gc,
                        double* gk) {
    HipEvaluator& ev = *static_cast<HipEvaluator*>(ctx);
    const int m = ev.D - ev.d;
    const size_t nV = (size_t)ev.D * m, nC = (size_t)m * m, nK = (size_t)ev.d * m, npar = nV + nC + nK;
    (void)k; (void)gv; (void)gc; (void)gk;                      // (contiguous behind v / cost: the driver's staging layout)
    const auto t0 = std::chrono::steady_clock::now();
    double* h_lam = const_cast<double*>(v) + P * npar;          // the staging buffer continues behind the P parameter sets
    double* h_vec = h_lam + P * m;
    for (int64_t p = 0; p < P && !ev.factors_staged; ++p) {
        const double* cp = c + p * nC;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) ev.a[i * m + j] = 0.5 * (cp[i * m + j] + cp[j * m + i]);
        if (!sym_eig(m, ev.a.data(), h_lam + p * m, ev.q.data(), ev.e.data())) return GABO_ERR_NOT_SPD;
        double* out = h_vec + p * nC;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) out[i * m + j] = ev.q[j * m + i];          // vectors into the columns
    }
    const auto t1 = std::chrono::steady_clock::now();
    double* in = ev.dev_in;
    double* out = ev.dev_out;
    if (hipMemcpyAsync(ev.dev_in, v, sizeof(double) * P * (npar + m + nC), hipMemcpyHostToDevice, ev.stream) != hipSuccess) return GABO_ERR_LAUNCH;
    double* dv = in;
    double* dc = dv + P * nV;
    double* dk = dc + P * nC;
    double* dlam = dk + P * nK;
    double* dvec = dlam + P * m;
    double* dcost = out;
    double* dgv = dcost + P;
    double* dgc = dgv + P * nV;
    double* dgk = dgc + P * nC;
    const int rc = nested_spd_reconstruction_launch(ev.data, ev.y, ev.sqrt_y, ev.w, dv, dc, dk, dcost, dgv, dgc, dgk, dlam, dvec, P, ev.N, ev.D, ev.d,
                                                    ev.metric, ev.recon_ws, ev.recon_ws_bytes, false, (gabo_stream_t)ev.stream);
    if (rc != GABO_OK) return rc;
    if (hipMemcpyAsync(cost, ev.dev_out, sizeof(double) * P * (1 + npar), hipMemcpyDeviceToHost, ev.stream) != hipSuccess) return GABO_ERR_LAUNCH;
    const auto t2 = std::chrono::steady_clock::now();
    const bool ok = hipStreamSynchronize(ev.stream) == hipSuccess;
    const auto t3 = std::chrono::steady_clock::now();
    ev.t_factor += std::chrono::duration<double>(t1 - t0).count();
    ev.t_enqueue += std::chrono::duration<double>(t2 - t1).count();
    ev.t_wait += std::chrono::duration<double>(t3 - t2).count();
    return ok ? GABO_OK : GABO_ERR_LAUNCH;
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace host
}  // namespace gabo

using namespace gabo::host;

extern "C" {

static int solve_impl(gabo_recon_eval_fn evaluate, void* ctx, const double* w_host, double* v, double* c, double* unit, double* raw, int D, int d,
                      double* staging, size_t staging_doubles, const gabo_recon_solve_options* options, gabo_recon_solve_log* log,
                      bool stage_factors) {
    if (D < 2 || D > GABO_SPD_MAX_DIM || d < 1 || d >= D) return GABO_ERR_DIM;
    if (!evaluate || !w_host || !v || !c || !unit || !raw || !staging || !options_ok(options)) return GABO_ERR_ARG;
    Dims dm(D, d);
    const size_t npar = (size_t)dm.nV + dm.nC + dm.nU;
    const size_t in_doubles = kMaxLookahead * (npar + dm.m + dm.nC);       // the parameter sets + room for an evaluator's factors of their C
    if (staging_doubles < in_doubles + kMaxLookahead * (1 + npar)) return GABO_ERR_ARG;
    Driver drv(D, d, w_host, evaluate, ctx, staging, staging + in_doubles, *options);
    drv.stage_factors = stage_factors;
    // host threads for the candidates of a line search (each candidate's retraction and factorisation is one task): default one per
    // candidate when the matrices are large enough for a task to outweigh the hand-off (~1 us) and the process may run on >= 16 cores (the
    // workers spin: on a small or busy machine they cost more than they bring - 8 shared vCPUs: 2.9 -> 3.6 ms per optimisation)
    int hw = (int)std::thread::hardware_concurrency();
#if defined(__gnu_linux__)
    cpu_set_t mask;                                    // the cores this process may actually run on (containers, taskset)
    if (sched_getaffinity(0, sizeof(mask), &mask) == 0) hw = std::min(hw, (int)CPU_COUNT(&mask));
#endif
    int threads = options->host_threads > 0 ? (int)options->host_threads : (dm.m >= 8 && hw >= 16 ? (int)kMaxLookahead : 1);
    threads = std::max(1, std::min(threads, (int)kMaxLookahead));
    StepPool pool(threads - 1);
    drv.pool = &pool;
    // default look-ahead: four step lengths per launch while the extra retractions are cheap - small matrices (measured single-threaded:
    // -22 % / -13 % / -3 % wall-clock at D = 5 / 10 / 20) or one thread per candidate
    drv.lookahead = options->lookahead < 1 ? ((dm.m <= 16 || threads >= 4) ? 4 : 2)
                                           : (options->lookahead > kMaxLookahead ? (int)kMaxLookahead : (int)options->lookahead);
    if (log) log->host_threads = threads;
    Point best(dm);
    std::memcpy(best.x.data(), v, sizeof(double) * dm.nV);
    std::memcpy(best.x.data() + dm.oC, c, sizeof(double) * dm.nC);
    std::memcpy(best.x.data() + dm.oU, unit, sizeof(double) * dm.nU);
    best.x[dm.oS] = *raw;
    drv.solve(best, log);
    std::memcpy(v, best.x.data(), sizeof(double) * dm.nV);
    std::memcpy(c, best.x.data() + dm.oC, sizeof(double) * dm.nC);
    std::memcpy(unit, best.x.data() + dm.oU, sizeof(double) * dm.nU);
    *raw = best.x[dm.oS];
    return drv.error;
}

int gabo_nested_spd_reconstruction_solve_with(gabo_recon_eval_fn evaluate, void* ctx, const double* w_host, double* v, double* c,
                                              double* unit, double* raw, int D, int d, double* staging, size_t staging_doubles,
                                              const gabo_recon_solve_options* options, gabo_recon_solve_log* log) {
    return solve_impl(evaluate, ctx, w_host, v, c, unit, raw, D, d, staging, staging_doubles, options, log, false);
}

void gabo_nested_spd_reconstruction_solve_workspace_bytes(int64_t N, int D, int d, size_t* device_bytes, size_t* pinned_doubles) {
    const size_t m = (size_t)(D - d), npar = (size_t)D * m + m * m + (size_t)d * m;
    const size_t in_doubles = kMaxLookahead * (npar + m + m * m);
    if (device_bytes)
        *device_bytes = align256(sizeof(double) * in_doubles) + align256(sizeof(double) * kMaxLookahead * (1 + npar)) +
                        align256(gabo_nested_spd_reconstruction_workspace_bytes(kMaxLookahead, N < 1 ? 1 : N, D, d));
    if (pinned_doubles) *pinned_doubles = in_doubles + kMaxLookahead * (1 + npar);
}

int gabo_nested_spd_reconstruction_solve(const double* data, const double* y, const double* sqrt_y, const double* w, const double* w_host,
                                         double* v, double* c, double* unit, double* raw, int64_t N, int D, int d, int metric,
                                         void* workspace, size_t workspace_bytes, double* pinned, size_t pinned_doubles,
                                         const gabo_recon_solve_options* options, gabo_recon_solve_log* log, gabo_stream_t stream) {
    if (D < 2 || D > GABO_SPD_MAX_DIM || d < 1 || d >= D) return GABO_ERR_DIM;
    if (!data || !y || !sqrt_y || !w || !workspace || !pinned || N < 0) return GABO_ERR_ARG;
    size_t need_dev = 0, need_pin = 0;
    gabo_nested_spd_reconstruction_solve_workspace_bytes(N, D, d, &need_dev, &need_pin);
    if (workspace_bytes < need_dev || pinned_doubles < need_pin) return GABO_ERR_ARG;
    const size_t m = (size_t)(D - d), npar = (size_t)D * m + m * m + (size_t)d * m;
    HipEvaluator e;
    e.data = data; e.y = y; e.sqrt_y = sqrt_y; e.w = w;
    e.N = N; e.D = D; e.d = d; e.metric = metric;
    char* base = static_cast<char*>(workspace);
    e.dev_in = reinterpret_cast<double*>(base);
    const size_t in_doubles = kMaxLookahead * (npar + m + m * m);
    e.dev_out = reinterpret_cast<double*>(base + align256(sizeof(double) * in_doubles));
    e.recon_ws = base + align256(sizeof(double) * in_doubles) + align256(sizeof(double) * kMaxLookahead * (1 + npar));
    e.a.resize(m * m); e.q.resize(m * m); e.lam.resize(m); e.e.resize(3 * m);
    // the tickets at the head of the launch workspace: cleared once, every launch leaves them at zero
    if (hipMemsetAsync(e.recon_ws, 0, 256, (hipStream_t)stream) != hipSuccess) return GABO_ERR_LAUNCH;
    e.recon_ws_bytes = gabo_nested_spd_reconstruction_workspace_bytes(kMaxLookahead, N < 1 ? 1 : N, D, d);
    e.stream = (hipStream_t)stream;
    e.factors_staged = true;
    const int rc = solve_impl(hip_evaluate, &e, w_host, v, c, unit, raw, D, d, pinned, pinned_doubles, options, log, true);
    if (getenv("GABO_RECON_TIMING") && log)
        fprintf(stderr, "gabo_nested_spd_reconstruction_solve D=%d: %.2f ms = manifold arithmetic %.2f + factor C %.2f + enqueue %.2f + wait %.2f (%ld launches)\n", D,
                1e3 * log->seconds, 1e3 * (log->seconds - log->seconds_evaluator), 1e3 * e.t_factor, 1e3 * e.t_enqueue, 1e3 * e.t_wait, (long)log->launches);
    return rc;
}

}  // extern "C"
