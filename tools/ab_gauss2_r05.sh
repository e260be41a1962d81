#!/bin/bash
# A/B of the d = 2 Gaussian Gram kernel variants (round 5), rocprofv3 kernel time of tools/config5_profile.py per library
cd /tmp && export TMPDIR=/tmp
for lib in "" ocml r32 ocml32; do
  if [ -n "$lib" ]; then export GABO_HIP_LIB=$GRAFT_REPO_ROOT/gabotorch_amd/libgabo_hip_$lib.so; else unset GABO_HIP_LIB; fi
  rm -rf /tmp/p5$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5$lib -o out -- python $GRAFT_REPO_ROOT/tools/config5_profile.py > /dev/null 2>&1
  f=$(find /tmp/p5$lib -name "out_kernel_stats.csv" | head -1)
  echo "== ${lib:-product}"; python $GRAFT_REPO_ROOT/tools/kstats.py $f 3
done
