// SPD pairwise Gram, dimensions 13..16: the same register-resident lane-per-pair kernels, two waves per SIMD up to d = 14, ONE above (512 VGPRs
// per lane); d = 17..20 in spd_pairwise_wide2.hip / spd_pairwise_wide3.hip.  Instantiations only; templates in spd_pairwise_body.hpp.
#include "spd_pairwise_body.hpp"

namespace gabo {

int launch_spd_ai_wide(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                       int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st) {
#define GABO_CASE(DD) \
    case DD:          \
        return launch_spd_ai<DD>(x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    switch (d) {
        GABO_CASE(13) GABO_CASE(14) GABO_CASE(15) GABO_CASE(16)
    }
#undef GABO_CASE
    if (d == 17 || d == 18) return launch_spd_ai_wide2(d, x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    if (d == 19 || d == 20) return launch_spd_ai_wide3(d, x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    return GABO_ERR_DIM;
}

}  // namespace gabo
