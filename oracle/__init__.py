"""CPU oracle for the manifold-kernel hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy/torch-CPU fp64, the arithmetic of the reference functions on the
hot path.  It exists to CHECK the HIP library, never to stand in for it:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `gabotorch_amd/` imports it, and the product path raises when the HIP library is absent.

Parity status: PINNED.  Every function here is checked by `tests/test_oracle_golden.py` against vectors
produced by importing the reference itself (`tests/golden/make_golden.py`, run in the development
container where /root/reference is mounted).  The pymanopt-side formulas (retr, inner,
egrad2rgrad, ehess2rhess, transp) have no reference test or source under /root/reference; those are pinned
only through the reference's own in-repo statements of the same maps (spd_utils.py:104-139,
sphere_utils.py:14-65, pymanopt_addons/tools/multi.py:55-75) and are marked "3P, unpinned" where no such
statement exists.
"""
