#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch9
mkdir -p $O
cd $R
for tag in main bwdjac; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 900 python tools/ab_backward.py $tag 2>&1 | grep -v amdgpu.ids >> $O/ab_backward.txt
done
cat $O/ab_backward.txt
timeout 600 python tools/sweep_bench.py 512 2>&1 | grep -v amdgpu.ids | tail -8 > $O/sweep.txt; cat $O/sweep.txt
timeout 600 python tools/tr_latency.py 2>&1 | grep -v amdgpu.ids | tail -12 > $O/tr_latency.txt; cat $O/tr_latency.txt
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
grep -E "^E  |passed|failed|rc=" $O/gputests.log | head -30
