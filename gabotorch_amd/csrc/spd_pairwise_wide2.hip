// SPD pairwise Gram, dimensions 17 and 18: the register-resident lane-per-pair kernels at one wave per SIMD (VGPRs + AGPRs, a little
// scratch from d = 18).  Instantiations only; templates in spd_pairwise_body.hpp.  Separate translation units so that the long
// compilations (30-40 s each) run in parallel.
#include "spd_pairwise_body.hpp"

namespace gabo {

int launch_spd_ai_wide2(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                        int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st) {
    if (d == 17) return launch_spd_ai<17>(x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    if (d == 18) return launch_spd_ai<18>(x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    return GABO_ERR_DIM;
}

}  // namespace gabo
