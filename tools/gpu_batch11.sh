#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_batch11
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value","ms_per_step","untimed_preheat_steps")}, "frac", l["roofline"]["frac"], "sphere", l["roofline_sphere"]["frac"], l["sphere_gram"]["ms_per_step"])
print("sweep", {k: v for k, v in l["acq_sweep"].items() if k.startswith("seconds")})
print("cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["all_threads"]["value"])
PY
GABO_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2-rank self-launch rc=$?"
tail -c 1500 $O/bench_2ranks.json | head -c 1500; echo; tail -3 $O/bench_2ranks.err
