// Stand-alone acquisition kernel, log-Euclidean surrogate (instantiations only).
#include "spd_acq_kernel.hpp"

namespace gabo {
int acq_log_euclidean(const AcqLaunch& a) { return dispatch_acq<1, 8>(a); }
}  // namespace gabo
