R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_nan.py tests/test_gpu_second_order.py -q 2>&1 | tail -30 > $O/tests.txt; cat $O/tests.txt
for i in 1 2; do
python tools/ab_sphere.py base 2>&1 | grep -v amdgpu.ids
for v in sp_every2 sp_every4 sp_every2w3 sp_every4w3 sp_chunks2 sp_chunks8 sp_probe1 sp_probe2; do
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_$v.so python tools/ab_sphere.py $v 2>&1 | grep -v amdgpu.ids
done
done | tee $O/ab_sphere.txt
