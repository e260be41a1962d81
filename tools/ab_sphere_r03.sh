cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k sphere 2>&1 | tail -2
for i in 1 2; do
for v in nopw g1p2; do GABO_HIP_LIB=gabotorch_amd/libgabo_hip_$v.so python tools/ab_sphere.py $v 2>&1 | grep -v amdgpu.ids; done
python tools/ab_sphere.py pw_g1 2>&1 | grep -v amdgpu.ids
done
python tools/ab_sphere.py pw_g1 8192 2>&1 | grep -v amdgpu.ids
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_nopw.so python tools/ab_sphere.py nopw 8192 2>&1 | grep -v amdgpu.ids
