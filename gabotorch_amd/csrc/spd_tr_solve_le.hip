// Single-launch trust-region solve, log-Euclidean surrogate (instantiations only; templates in spd_tr_body.hpp).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_log_euclidean(const SolveArgs& a) { return dispatch_solve<1>(a); }
}  // namespace gabo
