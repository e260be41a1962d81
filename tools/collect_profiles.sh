#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then separate PMC passes for HBM bytes.
# Outputs land in gpurun_out/profiles/ ; copy the summaries you want judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-symmetric > $OUT/bench_under_rocprof.json 2> /tmp/pk.err
cp /tmp/pk/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv
# PMC passes (counters only, no other trace domains): FETCH_SIZE and WRITE_SIZE cannot share a pass
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python $R/tools/prof_spd.py 4096 10 sym 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python $R/tools/prof_spd.py 4096 10 sym 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf2 -o f -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw2 -o w -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psph -o sph -- python $R/tools/prof_sphere.py > /dev/null 2>&1
cp /tmp/psph/sph_kernel_stats.csv $OUT/sphere_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psw -o sw -- python $R/tools/sweep_bench.py 512 > $OUT/sweep.log 2>&1
cp /tmp/psw/sw_kernel_stats.csv $OUT/sweep_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d /tmp/ps -o s -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
python - <<PY
import csv, collections, json
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "gabo" in name or "copy" in name.lower():
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
out = {}
for tag, path in (("sym_fetch", "/tmp/pf/f_counter_collection.csv"), ("sym_write", "/tmp/pw/w_counter_collection.csv"),
                  ("full_fetch", "/tmp/pf2/f_counter_collection.csv"), ("full_write", "/tmp/pw2/w_counter_collection.csv"),
                  ("full_sq", "/tmp/ps/s_counter_collection.csv")):
    try:
        a = load(path)
        out[tag] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in a.items()}
    except Exception as e:
        out[tag] = str(e)
json.dump(out, open("$OUT/pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
