R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_refresh
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout 900 python tools/hd_gabo_breakdown.py 2>&1 | grep -v amdgpu > $O/hd_gabo.txt
timeout 600 python tools/recon_native_probe.py 2>&1 | grep -v amdgpu > $O/recon_native.txt
timeout 600 python tools/hd_gabo_sphere_breakdown.py 2>&1 | grep -v amdgpu > $O/hd_gabo_sphere.txt
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/precn -o recn -- python $R/tools/recon_native_probe.py 20 > $O/recon_native.log 2>&1
cp /tmp/precn/recn_kernel_stats.csv $O/recon_native_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/phd -o hd -- python $R/tools/hd_gabo_breakdown.py --dims 20 --iters 4 > $O/hd_gabo.log 2>&1
cp /tmp/phd/hd_kernel_stats.csv $O/hd_gabo_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/phs -o hs -- python $R/tools/hd_gabo_sphere_breakdown.py 51 > $O/hd_gabo_sphere.log 2>&1
cp /tmp/phs/hs_kernel_stats.csv $O/hd_gabo_sphere_kernel_stats.csv
cd $R
cat $O/hd_gabo.txt $O/hd_gabo_sphere.txt $O/recon_native.txt; head -12 $O/recon_native_kernel_stats.csv | cut -c1-200; head -14 $O/hd_gabo_kernel_stats.csv | cut -c1-200
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(json.dumps({k: l[k] for k in ("value","ms_per_step")}), l["roofline"]["frac"], json.dumps(l["config5"]["latent_loop"]["reconstruction_optimisation"]))
PY
