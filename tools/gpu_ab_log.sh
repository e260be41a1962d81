#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for tag in main fdlog; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 600 python tools/ab_pairwise.py $tag 2>&1 | grep "SPD"
  GABO_HIP_LIB=$lib timeout 600 python tools/ab_small_d.py $tag 2>&1 | grep "SPD"
done
done
timeout 300 python tools/config5_bench.py 2>&1 | grep "config5 N"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
