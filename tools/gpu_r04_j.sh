R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
python tools/sweep_phases2.py 2>&1 | grep "device_rand" | tail -2
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_fpc.so python tools/sweep_phases2.py 2>&1 | grep "device_rand" | tail -2 | sed 's/^/[fpc] /'
done
GABO_HIP_LIB=$R/gabotorch_amd/libgabo_hip_fpc.so python -m pytest tests/test_gpu_optimize.py tests/test_gpu_ei_optimum.py tests/test_gpu_tr_traces.py -q 2>&1 | tail -8
