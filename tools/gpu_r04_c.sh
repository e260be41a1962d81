R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_nan.py -q -x -k "indefinite" 2>&1 | tail -40 > $O/nan.txt
python tools/pmc.py sphere_pairwise_kernel $O/pmc_sphere.json -- python $R/tools/prof_sphere.py > $O/pmc_sphere.log 2>&1
GABO_HIP_LIB=$R/gabotorch_amd/libgabo_hip_sp_probe2.so python tools/pmc.py sphere_pairwise_kernel $O/pmc_sphere_probe2.json -- python $R/tools/prof_sphere.py > $O/pmc_sphere_probe2.log 2>&1
cat $O/pmc_sphere.json
