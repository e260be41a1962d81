#!/bin/bash
export GABO_AB_DIMS=5,8,10
for rep in 1 2; do
for lib in bocml blog; do
  GABO_HIP_LIB=$GRAFT_REPO_ROOT/gabotorch_amd/libgabo_hip_$lib.so python tools/ab_backward.py $lib 2>&1 | grep "^\["
done
done
