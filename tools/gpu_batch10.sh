#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch10
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
grep -E "^E  |^FAILED|passed|failed|rc=" $O/gputests.log | head -30
