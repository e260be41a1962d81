// Single-launch trust-region solve, affine-invariant surrogate, d = 7, 8 (see spd_tr_solve.hip).
#include "spd_tr_body.hpp"

namespace gabo {
int solve_affine_invariant_hi(const SolveArgs& a) { return dispatch_solve<0, 7, 8>(a); }
}  // namespace gabo
