// Trust-region proposal kernels for the log-Euclidean surrogate, d = 7, 8 (see spd_tr_le.hip).
#include "spd_tr_body.hpp"

namespace gabo {
int propose_log_euclidean_hi(const ProposeArgs& a) { return dispatch_propose<1, 8, 7>(a); }
}  // namespace gabo
