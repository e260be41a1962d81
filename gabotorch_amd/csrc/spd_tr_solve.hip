// Single-launch trust-region solve, affine-invariant surrogate, d = 2 ... 6 (instantiations only; templates in spd_tr_body.hpp); d = 7, 8 in
// spd_tr_solve_hi.hip.
#include "spd_tr_body.hpp"

namespace gabo {
int solve_affine_invariant_hi(const SolveArgs& a);
bool solve_duo_wanted(const SolveArgs& a);                // spd_tr_solve_duo.hip: two waves per restart (latency regime, <= 512 restarts)
int solve_affine_invariant_duo(const SolveArgs& a);
int solve_affine_invariant(const SolveArgs& a) {
    if (solve_duo_wanted(a)) return solve_affine_invariant_duo(a);
    return a.d >= 7 ? solve_affine_invariant_hi(a) : dispatch_solve<0, 2, 6>(a);
}
}  // namespace gabo

#ifdef GABO_TR_CLOCKS
// development only: copy out and reset the (tag, cycle) pairs recorded by block 0 of the kernels of THIS translation unit
extern "C" int gabo_debug_clocks(long long* out, int max_pairs) {
    int n = 0;
    hipMemcpyFromSymbol(&n, HIP_SYMBOL(gabo_clk_n), sizeof(int));
    if (n > max_pairs) n = max_pairs;
    if (n > 4096) n = 4096;
    hipMemcpyFromSymbol(out, HIP_SYMBOL(gabo_clk_buf), (size_t)n * 2 * sizeof(long long));
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(gabo_clk_n), &zero, sizeof(int));
    return n;
}
#endif
