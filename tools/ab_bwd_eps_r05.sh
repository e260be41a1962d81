#!/bin/bash
export GABO_AB_DIMS=8,10,12
for rep in 1 2; do
for lib in bwd32 bwd26 bwd22; do
  GABO_HIP_LIB=$GRAFT_REPO_ROOT/gabotorch_amd/libgabo_hip_$lib.so python tools/ab_backward.py $lib 2>&1 | grep "^\["
done
done
