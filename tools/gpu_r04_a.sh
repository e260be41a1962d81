# round 4, first GPU pass: the new NaN / RCCL tests, the whole GPU suite, A/B of the sphere Gram with and without the NaN repair
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_nan.py -q -x 2>&1 | tail -15 > $O/nan_tests.txt; cat $O/nan_tests.txt
python -m pytest tests/test_gpu_multirank_bench.py -q -x 2>&1 | tail -15 > $O/multirank.txt; cat $O/multirank.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
for i in 1 2 3; do
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_nonanfix.so python tools/ab_sphere.py nonanfix 2>&1 | grep -v amdgpu.ids
python tools/ab_sphere.py nanfix 2>&1 | grep -v amdgpu.ids
done | tee $O/ab_sphere.txt
