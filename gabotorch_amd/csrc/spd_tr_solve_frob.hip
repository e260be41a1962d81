// Single-launch trust-region solve, Frobenius surrogate, d = 2 ... 8 (instantiations only; templates in spd_tr_body.hpp).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_frobenius(const SolveArgs& a) { return dispatch_solve<2, 2, 8>(a); }
}  // namespace gabo
