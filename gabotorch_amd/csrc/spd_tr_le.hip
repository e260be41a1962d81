// Trust-region proposal kernels for the log-Euclidean surrogate, d = 2 ... 6 (instantiations only; templates in spd_tr_body.hpp); d = 7, 8 in
// spd_tr_le_hi.hip.
#include "spd_tr_body.hpp"

namespace gabo {
int propose_log_euclidean_hi(const ProposeArgs& a);
int propose_log_euclidean(const ProposeArgs& a) { return a.d >= 7 ? propose_log_euclidean_hi(a) : dispatch_propose<1, 6>(a); }
}  // namespace gabo
