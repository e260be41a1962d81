#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch12
mkdir -p $O
cd $R
timeout 600 python tools/ab_pairwise.py main 2>&1 | grep -v amdgpu > $O/ab.txt; cat $O/ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value","ms_per_step","untimed_preheat_steps")}, "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"])
PY
timeout 600 python tools/dim_sweep.py > $O/dim_sweep.txt 2>&1; tail -20 $O/dim_sweep.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
