// Single-launch trust-region solve, log-Euclidean surrogate, d = 2 ... 5 (instantiations only; templates in spd_tr_body.hpp).  The other
// dimensions are in spd_tr_solve_le_mid.hip (6, 7) and spd_tr_solve_le_hi.hip (8): three translation units so that a clean build does not end on
// one 5-minute compilation (round 4: 271 s for all of them in this file).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_log_euclidean_mid(const SolveArgs& a);
int solve_log_euclidean_hi(const SolveArgs& a);
int solve_log_euclidean(const SolveArgs& a) {
    if (a.d >= 8) return solve_log_euclidean_hi(a);
    if (a.d >= 6) return solve_log_euclidean_mid(a);
    return dispatch_solve<1, 2, 5>(a);
}
}  // namespace gabo
