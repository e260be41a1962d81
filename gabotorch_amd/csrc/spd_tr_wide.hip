// Trust-region proposal kernels, affine-invariant surrogate, d = 9..12 (instantiations only; templates in spd_tr_body.hpp).
#include "spd_tr_body.hpp"

namespace gabo {
int propose_affine_invariant_wide(const ProposeArgs& a) { return dispatch_propose<0, 12, 9>(a); }
}  // namespace gabo
