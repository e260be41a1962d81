#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_reference_surface.py -x -q -m gpu > $O/surface.log 2>&1; echo "rc=$?" >> $O/surface.log
tail -30 $O/surface.log
for tag in main eps20 exsq; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 600 python tools/ab_pairwise.py $tag 2>&1 | grep -v amdgpu.ids | grep SPD >> $O/ab.txt
done
cat $O/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
tail -8 $O/gputests.log
