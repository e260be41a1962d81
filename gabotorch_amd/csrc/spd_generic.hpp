// Host entry points of the wave-per-pair fallback (spd_pairwise_generic.hip) used by the C ABI for 12 < d <= 32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gabo {
int launch_spd_ai_generic(const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1, int64_t n2,
                          int d, int64_t s1, int64_t s2, double beta, int flags, double* ws, int* status, hipStream_t st);
int launch_spd_ai_backward_generic(const double* x1, const double* x2, const double* gout, double* gx, int64_t batch, int64_t n1,
                                   int64_t n2, int d, int64_t s1, int64_t s2, int64_t go_sb, int64_t go_si, int64_t go_sj,
                                   double beta, int flags, double* ws, int* status, hipStream_t st);
}  // namespace gabo
