// Trust-region proposal kernels for the Frobenius surrogate, d = 2 ... 8 (instantiations only; templates in spd_tr_body.hpp).
#include "spd_tr_body.hpp"

namespace gabo {
int propose_frobenius(const ProposeArgs& a) { return dispatch_propose<2, 8>(a); }
}  // namespace gabo
