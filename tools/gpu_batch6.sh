#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_tr_traces.py -q -m gpu > $O/traces.log 2>&1; echo "rc=$?" >> $O/traces.log
grep -E "^E |passed|failed|rc=" $O/traces.log | head -40
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value","ms_per_step")}, l["roofline"]["frac"], l["roofline_sphere"]["frac"], l["sphere_gram"]["ms_per_step"])
print(json.dumps(l["config5"])[:1200])
print(json.dumps(l["cpu_baseline"])[:600])
print(json.dumps(l["acq_sweep"])[:900])
PY
