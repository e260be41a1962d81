// Trust-region proposal kernels for the log-Euclidean surrogate, d = 7, 8 (see spd_tr_le.hip; d = 8 only with GABO_LE_MAX_GENERIC_DIM >= 8,
// the default: spd_tr_body.hpp tells the story of that instantiation).
#include "spd_tr_body.hpp"

namespace gabo {
int propose_log_euclidean_hi(const ProposeArgs& a) { return dispatch_propose<1, (GABO_LE_MAX_GENERIC_DIM >= 8 ? 8 : 7), 7>(a); }
}  // namespace gabo
