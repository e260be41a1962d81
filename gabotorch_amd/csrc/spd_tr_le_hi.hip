// Trust-region proposal kernel for the log-Euclidean surrogate, d = 7 (see spd_tr_le.hip).  There is none for d = 8: that instantiation
// (generic workspace pointers, 512 registers and ~4 k spills) returned wrong proposals whatever its inputs - tools/soak_tr.py and
// tools/repro_solve_fault.py against the torch solver; the LDS-resident single-launch solve of the same size and the plan of tCG launches
// are right - so gabo_spd_tr_propose_supported says no and the caller takes one of those.
#include "spd_tr_body.hpp"

namespace gabo {
int propose_log_euclidean_hi(const ProposeArgs& a) { return dispatch_propose<1, 7, 7>(a); }
}  // namespace gabo
