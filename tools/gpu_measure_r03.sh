#!/bin/bash
# round-3 measurement pass: everything DESIGN.md section 5 quotes, in one run on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_measure
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python tools/dim_sweep.py 2>&1 | grep -v amdgpu > $O/dim_sweep.txt
timeout 600 python tools/tr_latency.py 2>&1 | grep -v amdgpu > $O/tr_latency.txt
timeout 900 python tools/config5_bench.py 2>&1 | grep -v amdgpu > $O/config5.txt
timeout 900 python tools/hd_gabo_breakdown.py 2>&1 | grep -v amdgpu > $O/hd_gabo.txt
timeout 600 python tools/recon_native_probe.py 2>&1 | grep -v amdgpu > $O/recon_native.txt
timeout 900 python tools/lds_eig_bench.py 2>&1 | grep -v amdgpu > $O/lds_eig.txt
timeout 900 python tools/gp_mll_bench.py 2>&1 | grep -v amdgpu > $O/gp_mll.txt
timeout 900 python tools/bo_iteration_breakdown.py 2>&1 | grep -v amdgpu > $O/bo_iteration.txt
timeout 600 python tools/sweep_bench.py 512 2>&1 | grep -v amdgpu > $O/sweep.txt
timeout 900 python tools/sphere_sweep_bench.py 2>&1 | grep -v amdgpu > $O/sphere_sweep.txt
(timeout 600 python tools/ab_sphere.py main; timeout 300 python tools/ab_solve.py main) 2>&1 | grep -v amdgpu > $O/sphere.txt
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DGABO_EIGH_CLOCKS -I gabotorch_amd/csrc tools/ubench_eigh.hip -o /tmp/ubench_eigh 2> /dev/null && /tmp/ubench_eigh > $O/ubench_eigh.txt
for f in dim_sweep tr_latency config5 hd_gabo recon_native lds_eig gp_mll bo_iteration sweep sphere_sweep sphere ubench_eigh; do echo "== $f"; tail -25 $O/$f.txt | cut -c1-300; done
python - <<PY
import json
l=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(json.dumps({k: l[k] for k in ("value","ms_per_step")}), l["roofline"]["frac"], l["symmetric_gram"]["ms_per_step"], l["sphere_gram"]["ms_per_step"], l["surrogate_fit"]["ms"], l["acq_sweep_sphere"]["seconds"], l["config5"]["latent_loop"]["reconstruction_evaluation_ms"])
PY
