#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for tag in main bnot; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_AB_DIMS=10 GABO_HIP_LIB=$lib timeout 300 python tools/ab_pairwise.py $tag 2>&1 | grep "d=10"
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py -q 2>&1 | tail -2
