#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch5
mkdir -p $O
cd $R
for tag in main nudge la2; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 600 python tools/ab_pairwise.py $tag 2>&1 | grep -v amdgpu.ids | grep SPD >> $O/ab.txt
done
cat $O/ab.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
tail -8 $O/gputests.log
