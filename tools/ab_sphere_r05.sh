#!/bin/bash
# sphere Gram N = 4096, S^9: kernel time with the cached kernel-value table (product) under rocprofv3, 400 launches from a cold start
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/psph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psph -o out -- python $GRAFT_REPO_ROOT/tools/prof_sphere.py 4096 400 > /dev/null 2>&1
f=$(find /tmp/psph -name "out_kernel_stats.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/kstats.py $f 3
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05h; cp $f $GRAFT_REPO_ROOT/gpurun_out/r05h/sphere_kernel_stats.csv
