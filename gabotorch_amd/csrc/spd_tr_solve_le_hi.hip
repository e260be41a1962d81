// Single-launch trust-region solve, log-Euclidean surrogate, d = 8 (see spd_tr_solve_le.hip).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_log_euclidean_hi(const SolveArgs& a) { return dispatch_solve<1, 8, 8>(a); }
}  // namespace gabo
