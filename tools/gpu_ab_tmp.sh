#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for tag in main rawsq; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_AB_DIMS=10 GABO_HIP_LIB=$lib timeout 300 python tools/ab_pairwise.py $tag 2>&1 | grep "d=10"
done
done
