#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch8
mkdir -p $O
cd $R
timeout 900 python tools/ab_backward.py main 2>&1 | grep -v amdgpu.ids > $O/ab_backward.txt; cat $O/ab_backward.txt
timeout 1800 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_tr_traces.py tests/test_gpu_configs_full_size.py -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -E "^E  |passed|failed|rc=" $O/tests.log | head -30
