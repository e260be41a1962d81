#!/bin/bash
# round-2 GPU batch 1: parity of the changed kernels, MFMA micro-benchmark, pairwise A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -5 $O/parity.log
hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_mfma_f64.hip -o /tmp/ubm 2>/dev/null && timeout 120 /tmp/ubm > $O/ubench_mfma_f64.txt 2>&1
cat $O/ubench_mfma_f64.txt
hipcc -O3 --offload-arch=gfx950 tools/ubench_fp64.hip -o /tmp/ubf 2>/dev/null && timeout 120 /tmp/ubf > $O/ubench_fp64.txt 2>&1
tail -4 $O/ubench_fp64.txt
for tag in main noql nr1; do
  lib=$R/gabotorch_amd/libgabo_hip.so; [ $tag != main ] && lib=$R/gabotorch_amd/libgabo_hip_$tag.so
  GABO_HIP_LIB=$lib timeout 600 python tools/ab_pairwise.py $tag >> $O/ab.txt 2>&1
done
cat $O/ab.txt
