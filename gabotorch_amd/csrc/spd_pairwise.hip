// SPD affine-invariant pairwise kernel matrix on gfx950.
//
//   d_ij^2 = sum_k log^2 lambda_k(L_i^-1 B_j L_i^-T) + 1e-15,   L_i = chol(A_i)
//
// replaces kernel_utils/kernels_spd.py:72-100 + Riemannian_utils/spd_utils_torch.py:53-121,159-194.
//
// Two launches:
//   1. spd_prep_kernel  - one lane per input matrix: Mandel vector -> Cholesky.  x1 side stores L^-1 (packed lower,
//      row contiguous: read back through the scalar cache, it is wave-uniform in step 2); x2 side stores the factor
//      G (B = G G^T) entry-major ("SoA", [tri][n2]) so that lane j reads G[e][j] fully coalesced.
//   2. spd_ai_pairwise_kernel - one LANE per pair, a wave = 64 consecutive columns j of one row i at a time.
//      C = L_i^-1 G_j (lower x lower), M = C C^T (symmetric, lower kept), eigenvalues of M by the per-lane
//      tridiagonal/QL solver of spd_eig.hpp, all in registers.  No LDS, no cross-lane traffic; the 1.8 MB operand
//      sets stay L2 resident, the only HBM stream is the N1 x N2 output.
#include "spd_pairwise_body.hpp"


namespace gabo {
int launch_spd_ai_prepared(int d, double* out, int64_t batch, int64_t n1, int64_t n2, bool shared1, bool shared2, double beta, int flags,
                           double* ws, hipStream_t st) {
    const int64_t s1 = shared1 ? 0 : 1, s2 = shared2 ? 0 : 1;
    switch (d) {
        case 2: return launch_spd_ai<2>(nullptr, nullptr, out, nullptr, batch, n1, n2, s1, s2, beta, flags, ws, nullptr, st, true);
        case 3: return launch_spd_ai<3>(nullptr, nullptr, out, nullptr, batch, n1, n2, s1, s2, beta, flags, ws, nullptr, st, true);
        case 4: return launch_spd_ai<4>(nullptr, nullptr, out, nullptr, batch, n1, n2, s1, s2, beta, flags, ws, nullptr, st, true);
    }
    return GABO_ERR_DIM;
}
}  // namespace gabo

extern "C" {

size_t gabo_spd_ai_workspace_bytes(int64_t batch, int64_t n1, int64_t n2, int d) {
    if (batch < 0 || n1 < 0 || n2 < 0 || d < 1) return 0;
    const size_t packed = (size_t)(batch * (n1 + n2)) * (size_t)gabo::tri_size(d) * sizeof(double);
    // wave-per-pair fallback (forward above GABO_SPD_FWD_REG_MAX_DIM, backward above GABO_SPD_REG_MAX_DIM): L^-1 (d x d) per x1
    // matrix, and as much again for the backward sums
    const size_t generic = (size_t)(2 * batch * n1) * (size_t)d * (size_t)d * sizeof(double);
    if (d > GABO_SPD_FWD_REG_MAX_DIM) return generic;
    if (d > GABO_SPD_REG_MAX_DIM) return packed > generic ? packed : generic;
    return packed;
}

int gabo_spd_ai_pairwise(const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2, int d,
                         int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags, void* workspace,
                         size_t workspace_bytes, int* status, gabo_stream_t stream) {
    if (batch < 0 || n1 < 0 || n2 < 0 || x1_batch_stride < 0 || x2_batch_stride < 0) return GABO_ERR_ARG;
    if (d < 2 || d > GABO_SPD_MAX_DIM) return GABO_ERR_DIM;
    if (batch == 0 || n1 == 0 || n2 == 0) return GABO_OK;
    if (!x1 || !x2 || !out || !workspace || !status) return GABO_ERR_ARG;
    if (workspace_bytes < gabo_spd_ai_workspace_bytes(batch, n1, n2, d)) return GABO_ERR_ARG;
    if ((flags & GABO_SYMMETRIC) && (n1 != n2 || batch > 65535)) return GABO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    if (d > GABO_SPD_FWD_REG_MAX_DIM)
        return gabo::launch_spd_ai_generic(x1, x2, out, dist_out, batch, n1, n2, d, x1_batch_stride, x2_batch_stride, beta, flags, ws,
                                           status, st);
#define GABO_CASE(DD) \
    case DD:          \
        return gabo::launch_spd_ai<DD>(x1, x2, out, dist_out, batch, n1, n2, x1_batch_stride, x2_batch_stride, beta, flags, ws, status, st);
    switch (d) {
#ifdef GABO_ONLY_DIM  /* development builds: one instantiation compiles in seconds */
        GABO_CASE(GABO_ONLY_DIM)
#else
        GABO_CASE(2) GABO_CASE(3) GABO_CASE(4) GABO_CASE(5) GABO_CASE(6) GABO_CASE(7) GABO_CASE(8) GABO_CASE(9) GABO_CASE(10)
        GABO_CASE(11) GABO_CASE(12)
#endif
    }
#undef GABO_CASE
#ifndef GABO_ONLY_DIM
    if (d > GABO_SPD_REG_MAX_DIM)
        return gabo::launch_spd_ai_wide(d, x1, x2, out, dist_out, batch, n1, n2, x1_batch_stride, x2_batch_stride, beta, flags, ws, status, st);
#endif
    return GABO_ERR_DIM;
}

}  // extern "C"
