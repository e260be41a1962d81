// SPD pairwise Gram, dimensions 19 and 20 (see spd_pairwise_wide2.hip).
#include "spd_pairwise_body.hpp"

namespace gabo {

int launch_spd_ai_wide3(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                        int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st) {
    if (d == 19) return launch_spd_ai<19>(x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    if (d == 20) return launch_spd_ai<20>(x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    return GABO_ERR_DIM;
}

}  // namespace gabo
