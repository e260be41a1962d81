// Trust-region proposal kernels for the log-Euclidean surrogate (instantiations only; templates in spd_tr_body.hpp).
#include "spd_tr_body.hpp"

namespace gabo {
int propose_log_euclidean(const ProposeArgs& a) { return dispatch_propose<1, 8>(a); }
}  // namespace gabo
