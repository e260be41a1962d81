#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch3
mkdir -p $O
cd $R
for tag in main sph_c1 sph_c1nc sph_c2 sph_c4nc sph_p1 sph_p2 sph_p1c1; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 300 python tools/ab_sphere.py $tag 2>&1 | grep -v amdgpu.ids >> $O/ab_sphere.txt
done
cat $O/ab_sphere.txt
