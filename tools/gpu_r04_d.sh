R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_nan.py tests/test_gpu_configs_full_size.py -q -k "sphere" 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
for i in 1 2 3; do
python tools/ab_sphere.py kt2 2>&1 | grep -v amdgpu.ids
for v in sp_nokt sp_kt1 sp_kt4; do GABO_HIP_LIB=gabotorch_amd/libgabo_hip_$v.so python tools/ab_sphere.py $v 2>&1 | grep -v amdgpu.ids; done
done | tee $O/ab_sphere.txt
python tools/ab_sphere.py kt2 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_sphere.txt
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_sp_nokt.so python tools/ab_sphere.py sp_nokt 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_sphere.txt
python tools/pmc.py sphere_pairwise_kernel $O/pmc_sphere_kt.json -- python $R/tools/prof_sphere.py > $O/pmc_sphere_kt.log 2>&1
