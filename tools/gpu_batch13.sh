#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch13
mkdir -p $O
cd $R
for rep in 1 2; do
for tag in main fin0 fin2 fin3; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 600 python tools/ab_pairwise.py $tag 2>&1 | grep "SPD d=10" >> $O/ab.txt
done
done
cat $O/ab.txt
