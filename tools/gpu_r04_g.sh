R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_g
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gpu_suite.txt
for n in 2048 4096 8192 16384; do python tools/ab_sphere.py main $n 2>&1 | grep -v amdgpu.ids; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psph -o sph -- python $R/tools/prof_sphere.py > /dev/null 2>&1
cp /tmp/psph/*/sph_kernel_stats.csv $O/sphere_kernel_stats.csv 2>/dev/null || cp /tmp/psph/sph_kernel_stats.csv $O/sphere_kernel_stats.csv
head -5 $O/sphere_kernel_stats.csv | cut -c1-220
