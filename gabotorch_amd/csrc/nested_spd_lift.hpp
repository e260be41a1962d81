// Extreme eigenpairs of a latent SPD point lifted to the original space, one wave per point: the device body shared by the standalone
// launch (nested_spd_constraints.hip) and the single-launch trust-region solve (spd_tr_body.hpp).  See nested_spd_constraints.hip.
#pragma once
#include "gabo_device.hpp"
#include "lds_linalg.hpp"

namespace gabo {

// LDS doubles the body needs for an original dimension D and a latent dimension d
__host__ __device__ inline size_t nested_extremes_lds_doubles(int D, int d) {
    return (size_t)2 * D * D + (size_t)4 * D * d + (size_t)6 * d * d + (size_t)4 * d + kJacobiScratch;
}

// y: the latent point (d x d, global or LDS); w, p (D x d), x0 (D x D): gabo_nested_spd_lift_prepare.  lds: nested_extremes_lds_doubles(D, d).
// Out (LDS, valid after the closing barrier): lam_out[0..1] = lambda_max, lambda_min of Xrec(y); grad_out (when want_grad): 2 d^2 doubles,
// their gradients with respect to the symmetric y.  Returns pointers into `lds`.  Block = one wave (64 threads).
struct NestedExtremesOut {
    const double* lam;
    const double* grad;
};

// MAXDP: see wave_eig_extremes_any (D <= MAXDP is the caller's to check)
template <bool QLD, bool QLd, int MAXDP = 32>
__device__ __forceinline__ NestedExtremesOut nested_extremes_body(const double* __restrict__ y, const double* __restrict__ w,
                                                                  const double* __restrict__ p, const double* __restrict__ x0, int D, int d,
                                                                  double* lds, bool want_grad) {
    const int DD = D * D, dd = d * d;
    double* M0 = lds;                 // D x D
    double* M1 = M0 + DD;             // D x D scratch of the eigen-solver
    double* Wl = M1 + DD;             // D x d
    double* Pl = Wl + D * d;          // D x d
    double* WY = Pl + D * d;          // D x d
    double* WS = WY + D * d;          // D x d
    double* Yl = WS + D * d;          // d x d
    double* Lm = Yl + dd;             // d x d : eigenvalues of Y on the diagonal
    double* Us = Lm + dd;             // d x d
    double* Sm = Us + dd;             // d x d : Y^1/2
    double* G1 = Sm + dd;             // d x d
    double* G2 = G1 + dd;             // d x d
    double* ab = G2 + dd;             // 4 d : a_max, b_max, a_min, b_min
    double* cs = ab + 4 * d;
    for (int e = threadIdx.x; e < D * d; e += 64) {
        Wl[e] = w[e];
        Pl[e] = p[e];
    }
    for (int e = threadIdx.x; e < dd; e += 64) {
        const int r = e / d, cc = e - r * d;
        Yl[e] = 0.5 * (y[r * d + cc] + y[cc * d + r]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < dd; e += 64) Lm[e] = Yl[e];
    __syncthreads();
    lds_eigh<QLd>(Lm, Us, cs, d);
    lds_fun_from_eig(Lm, Us, Sm, d, FN_SQRT, cs);
    for (int e = threadIdx.x; e < D * d; e += 64) {
        const int r = e / d, a = e - r * d;
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < d; ++q) {
            s1 = __builtin_fma(Wl[r * d + q], Yl[q * d + a], s1);
            s2 = __builtin_fma(Wl[r * d + q], Sm[q * d + a], s2);
        }
        WY[e] = s1;
        WS[e] = s2;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < DD; e += 64) {
        const int r = e / D, cc = e - r * D;
        if (cc > r) continue;
        double s = x0[r * D + cc];
        for (int a = 0; a < d; ++a)
            s = __builtin_fma(WY[r * d + a], Wl[cc * d + a], __builtin_fma(WS[r * d + a], Pl[cc * d + a], __builtin_fma(Pl[r * d + a], WS[cc * d + a], s)));
        M0[r * D + cc] = s;
        M0[cc * D + r] = s;
    }
    __syncthreads();
    if constexpr (QLD) {
        wave_eig_extremes_any<MAXDP>(M0, M1, cs, D);     // M0[0..D) = v_max, M0[D..2D) = v_min, cs[0..1] = lambda_max, lambda_min
        __syncthreads();
    } else {                                             // D < 5: the Jacobi solver, then pick
        lds_jacobi(M0, M1, cs, D);
        int imax = 0, imin = 0;
        for (int q = 1; q < D; ++q) {
            if (M0[q * D + q] > M0[imax * D + imax]) imax = q;
            if (M0[q * D + q] < M0[imin * D + imin]) imin = q;
        }
        const double lmax = M0[imax * D + imax], lmin = M0[imin * D + imin];
        __syncthreads();
        if (threadIdx.x < D) {
            M0[threadIdx.x] = M1[threadIdx.x * D + imax];
            M0[D + threadIdx.x] = M1[threadIdx.x * D + imin];
        }
        if (threadIdx.x == 0) {
            cs[0] = lmax;
            cs[1] = lmin;
        }
        __syncthreads();
    }
    NestedExtremesOut out{cs, M1};
    if (!want_grad) return out;
    // a = W^T v, b = P^T v for both vectors
    for (int e = threadIdx.x; e < 4 * d; e += 64) {
        const int h = e / (2 * d), rem = e - h * 2 * d, which = rem / d, a = rem - which * d;
        const double* src = which ? Pl : Wl;
        double s = 0.0;
        for (int q = 0; q < D; ++q) s = __builtin_fma(src[q * d + a], M0[h * D + q], s);
        ab[e] = s;
    }
    __syncthreads();
    for (int h = 0; h < 2; ++h) {
        const double* av = ab + h * 2 * d;
        const double* bv = av + d;
        // G1 = Us^T (a b^T + b a^T) Us o 1 / (sqrt mu_k + sqrt mu_l)
        for (int e = threadIdx.x; e < dd; e += 64) {
            const int r = e / d, cc = e - r * d;
            double ua = 0.0, ub = 0.0, va = 0.0, vb = 0.0;     // (Us^T a)_r, (Us^T b)_r, (Us^T a)_c, (Us^T b)_c
            for (int q = 0; q < d; ++q) {
                ua = __builtin_fma(Us[q * d + r], av[q], ua);
                ub = __builtin_fma(Us[q * d + r], bv[q], ub);
                va = __builtin_fma(Us[q * d + cc], av[q], va);
                vb = __builtin_fma(Us[q * d + cc], bv[q], vb);
            }
            G1[e] = (ua * vb + ub * va) / (__builtin_sqrt(Lm[r * d + r]) + __builtin_sqrt(Lm[cc * d + cc]));
        }
        __syncthreads();
        lds_mm(Us, G1, G2, d, false, false);
        lds_mm(G2, Us, G1, d, false, true);
        for (int e = threadIdx.x; e < dd; e += 64) {
            const int r = e / d, cc = e - r * d;
            M1[h * dd + e] = __builtin_fma(av[r], av[cc], 0.5 * (G1[r * d + cc] + G1[cc * d + r]));     // (M1 is free after the eigen-solve)
        }
        __syncthreads();
    }
    return out;
}

}  // namespace gabo
