// Single-launch trust-region solve, log-Euclidean surrogate, d = 6, 7 (see spd_tr_solve_le.hip).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_log_euclidean_mid(const SolveArgs& a) { return dispatch_solve<1, 6, 7>(a); }
}  // namespace gabo
