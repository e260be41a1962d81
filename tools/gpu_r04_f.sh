R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_f
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_nan.py -q -k "sphere" 2>&1 | tail -5
for i in 1 2 3; do
python tools/ab_sphere.py kt1024 2>&1 | grep -v amdgpu.ids
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_kt512.so python tools/ab_sphere.py kt512 2>&1 | grep -v amdgpu.ids
done | tee $O/ab.txt
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_clk_kt.so python tools/sphere_clocks.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
