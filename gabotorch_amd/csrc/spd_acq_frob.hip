// Stand-alone acquisition kernel, Frobenius surrogate (kernels_spd.py:190-241; instantiations only: the evaluation is acq_eval_frob<D, 2> of
// spd_acq_body.hpp - the candidate's Mandel vector is its own feature vector, no eigen-decomposition).
#include "spd_acq_kernel.hpp"

namespace gabo {
int acq_frobenius(const AcqLaunch& a) { return dispatch_acq<2, 8>(a); }
}  // namespace gabo
