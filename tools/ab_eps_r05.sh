#!/bin/bash
# the three threshold builds interleaved twice on one box
for rep in 1 2; do
for lib in eps20 eps16 eps14; do
  GABO_HIP_LIB=$GRAFT_REPO_ROOT/gabotorch_amd/libgabo_hip_$lib.so python tools/ab_eps_gauss.py $lib 2>&1 | tail -1
done
done
