// SPD pairwise Gram, dimensions 17..20 (instantiations only; see spd_pairwise_wide.hip).
#include "spd_pairwise_body.hpp"

namespace gabo {

int launch_spd_ai_wide2(int d, const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                        int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st) {
#define GABO_CASE(DD) \
    case DD:          \
        return launch_spd_ai<DD>(x1, x2, out, dist_out, batch, n1, n2, s1, s2, beta, flags, ws, status, st);
    switch (d) {
        GABO_CASE(17) GABO_CASE(18) GABO_CASE(19) GABO_CASE(20)
    }
#undef GABO_CASE
    return GABO_ERR_DIM;
}

}  // namespace gabo
