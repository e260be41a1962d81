#!/bin/bash
for rep in 1 2; do
for lib in base r4 r16 t128; do
  GABO_HIP_LIB=$GRAFT_REPO_ROOT/gabotorch_amd/libgabo_hip_$lib.so python tools/ab_eps_gauss.py $lib 2>&1 | tail -1 | cut -c1-40
done
done
