#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then SEPARATE PMC passes (counters only, no other trace
# domains) for HBM bytes and instruction counts of the three kernels the round worked on.  Raw outputs land in
# gpurun_out/profiles_r05/ ; tools/summarise_profiles.py r05 turns them into the files committed under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench command itself under the kernel trace (same flags as the graded run minus the CPU baseline and the side reports)
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-symmetric > $OUT/bench_under_rocprof.json 2> /tmp/pk.err
cp /tmp/pk/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv
cp /tmp/pk/bench_kernel_trace.csv $OUT/bench_kernel_trace.csv
# 2. PMC passes on the Gram launch (FETCH_SIZE and WRITE_SIZE cannot share a pass; the mirror kernel of the symmetric build is the
#    known-byte-count calibration of FETCH_SIZE)
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python $R/tools/prof_spd.py 4096 10 sym 2 > /dev/null 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf2 -o f -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw2 -o w -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d /tmp/ps -o s -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
# 3. sphere Gram: kernel trace + instruction counters + write bytes
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psph -o sph -- python $R/tools/prof_sphere.py 4096 400 > /dev/null 2>&1
cp /tmp/psph/sph_kernel_stats.csv $OUT/sphere_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d /tmp/pss -o s -- python $R/tools/prof_sphere.py > /dev/null 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/psw -o w -- python $R/tools/prof_sphere.py > /dev/null 2>&1
# 3b. the round-4 instruction / stall / LDS counters of the sphere Gram, the headline kernel and the trust-region solve (tools/pmc.py)
python $R/tools/pmc.py sphere_pairwise_kernel $OUT/pmc_sphere.json -- python $R/tools/prof_sphere.py > /dev/null 2>&1
python $R/tools/pmc.py spd_ai_pairwise_kernel $OUT/pmc_headline.json -- python $R/tools/prof_spd.py 4096 10 x 3 > /dev/null 2>&1
python $R/tools/pmc.py spd_tr_solve_kernel $OUT/pmc_tr_solve.json -- python $R/tools/sweep_once.py > /dev/null 2>&1
# 4. backward (d = 10, N = 4096) and the config-4 sweep under the kernel trace
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/tools/ab_backward.py prof > $OUT/backward.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $OUT/backward_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psweep -o sw -- python $R/tools/sweep_bench.py 512 > $OUT/sweep.log 2>&1
cp /tmp/psweep/sw_kernel_stats.csv $OUT/sweep_kernel_stats.csv
# 4b. config 5's latent loop: the reconstruction optimiser (fused launch) and the HD-GaBO example under the kernel trace; the tiled likelihood
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prec -o rec -- python $R/tools/recon_profile.py > $OUT/recon.log 2>&1
cp /tmp/prec/rec_kernel_stats.csv $OUT/recon_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/precn -o recn -- python $R/tools/recon_native_probe.py 20 > $OUT/recon_native.log 2>&1
cp /tmp/precn/recn_kernel_stats.csv $OUT/recon_native_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/phd -o hd -- python $R/tools/hd_gabo_breakdown.py --dims 20 --iters 4 > $OUT/hd_gabo.log 2>&1
cp /tmp/phd/hd_kernel_stats.csv $OUT/hd_gabo_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmll -o mll -- python $R/tools/gp_mll_bench.py > $OUT/gp_mll.log 2>&1
cp /tmp/pmll/mll_kernel_stats.csv $OUT/gp_mll_kernel_stats.csv
# 5. the config-5 pieces (projection, nested Gram, logm, log-Euclidean Gram) under the kernel trace, then their instruction counters and write bytes
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -o c5 -- python $R/tools/config5_profile.py > /dev/null 2>&1
cp /tmp/pc5/c5_kernel_stats.csv $OUT/config5_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_LDS --output-format csv -d /tmp/pc5s -o s -- python $R/tools/config5_profile.py > /dev/null 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pc5w -o w -- python $R/tools/config5_profile.py > /dev/null 2>&1
python - <<PY
import csv, collections, json
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "gabo" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
out = {}
for tag, path in (("sym_fetch", "/tmp/pf/f_counter_collection.csv"), ("full_fetch", "/tmp/pf2/f_counter_collection.csv"),
                  ("full_write", "/tmp/pw2/w_counter_collection.csv"), ("full_sq", "/tmp/ps/s_counter_collection.csv"),
                  ("sphere_sq", "/tmp/pss/s_counter_collection.csv"), ("sphere_write", "/tmp/psw/w_counter_collection.csv"),
                  ("config5_sq", "/tmp/pc5s/s_counter_collection.csv"), ("config5_write", "/tmp/pc5w/w_counter_collection.csv")):
    try:
        a = load(path)
        out[tag] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in a.items()}
    except Exception as e:
        out[tag] = str(e)
json.dump(out, open("$OUT/pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
