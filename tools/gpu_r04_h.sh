R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_h
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_nested_gram.py tests/test_gpu_parity.py -q -x 2>&1 | tail -15 | tee $O/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(json.dumps({k: l[k] for k in ("value","ms_per_step","backend","world_size")}), l["roofline"]["frac"])
print(json.dumps(l["roofline_sphere"]))
print(json.dumps(l["config5"]["device_ms_in_a_hip_graph"]))
print(json.dumps(l["cpu_baseline"])[:600])
PY
