R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_e
mkdir -p $O
cd $R
for v in clk_kt clk_kt_p2 clk_pw; do echo "== $v"; GABO_HIP_LIB=gabotorch_amd/libgabo_hip_$v.so python tools/sphere_clocks.py 2>&1 | grep -v amdgpu.ids; done | tee $O/clocks.txt
GABO_HIP_LIB=gabotorch_amd/libgabo_hip_kt_p2.so python tools/ab_sphere.py kt_probe2 2>&1 | grep -v amdgpu.ids | tee -a $O/clocks.txt
