#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch7
mkdir -p $O
cd $R
timeout 600 python tools/ab_pairwise.py main 2>&1 | grep SPD > $O/ab.txt; cat $O/ab.txt
timeout 1800 python -m pytest tests/test_gpu_tr_traces.py tests/test_gpu_configs_full_size.py -q -m gpu --durations=8 > $O/newtests.log 2>&1; echo "rc=$?" >> $O/newtests.log
grep -E "^E |passed|failed|rc=|s call" $O/newtests.log | head -40
