# round 5: GPU test suite, bench line, rocprofv3 / PMC collection (tools/collect_profiles_r05.sh), HD-GaBO breakdowns
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_refresh
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/gpu_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/collect_profiles_r05.sh > $O/collect.log 2>&1
timeout 900 python tools/hd_gabo_breakdown.py 2>&1 | grep -v amdgpu > $O/hd_gabo.txt
timeout 600 python tools/hd_gabo_sphere_breakdown.py 2>&1 | grep -v amdgpu > $O/hd_gabo_sphere.txt
tail -5 $O/hd_gabo.txt $O/hd_gabo_sphere.txt
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(json.dumps({k: l[k] for k in ("value","ms_per_step")}), l["roofline"]["frac"], json.dumps(l["roofline_sphere"])[:300])
print(json.dumps(l["acq_sweep"])[:600])
PY
