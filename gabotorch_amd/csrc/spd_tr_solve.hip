// Single-launch trust-region solve, affine-invariant surrogate (instantiations only; templates in spd_tr_body.hpp).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_affine_invariant(const SolveArgs& a) { return dispatch_solve<0>(a); }
}  // namespace gabo
