#!/bin/bash
# Runs on the GPU box (via gpurun): the round-6 evidence behind profiles/r06_*.  rocprofv3 kernel-trace stats of the bench command, PMC passes (counters
# only, no other trace domains) of the headline kernel, the backward and the solve kernel, the sweep under the kernel trace (its three launches), the
# host timeline of a sweep.  Raw outputs land in gpurun_out/profiles_r06/ ; the files worth keeping are copied to profiles/ by hand (small, named r06_*).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench command under the kernel trace (same flags as the graded run minus the CPU baseline and the side reports)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-symmetric --no-plugin-surface --traffic file > $OUT/bench_under_rocprof.json 2> /tmp/pk.err
cp /tmp/pk/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv
# 2. instruction / stall counters of the three kernels the round worked on or is judged on (tools/pmc.py: one rocprofv3 run per counter set)
python $R/tools/pmc.py spd_ai_pairwise_kernel $OUT/pmc_headline.json -- python $R/tools/prof_spd.py 4096 10 x 3 > /dev/null 2>&1
GABO_AB_DIMS=10 python $R/tools/pmc.py spd_ai_backward_kernel $OUT/pmc_backward.json -- python $R/tools/ab_backward.py prof > /dev/null 2>&1
python $R/tools/pmc.py spd_tr_solve_duo_kernel $OUT/pmc_tr_solve_two_waves.json -- python $R/tools/sweep_once.py 512 > /dev/null 2>&1
GABO_TR_DUO=0 python $R/tools/pmc.py spd_tr_solve_kernel $OUT/pmc_tr_solve.json -- python $R/tools/sweep_once.py 512 > /dev/null 2>&1
# 2b. HBM bytes of the headline launch (FETCH_SIZE and WRITE_SIZE cannot share a pass)
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf2 -o f -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw2 -o w -- python $R/tools/prof_spd.py 4096 10 x 2 > /dev/null 2>&1
cp /tmp/pf2/f_counter_collection.csv $OUT/headline_fetch_size.csv 2>/dev/null
cp /tmp/pw2/w_counter_collection.csv $OUT/headline_write_size.csv 2>/dev/null
# 3. the sweeps (64 and 512 restarts) and the backward under the kernel trace
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps64 -o sw -- python $R/tools/sweep_once.py 64 > /dev/null 2>&1
cp /tmp/ps64/sw_kernel_stats.csv $OUT/sweep64_kernel_stats.csv; cp /tmp/ps64/sw_kernel_trace.csv $OUT/sweep64_kernel_trace.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps512 -o sw -- python $R/tools/sweep_once.py 512 > /dev/null 2>&1
cp /tmp/ps512/sw_kernel_stats.csv $OUT/sweep512_kernel_stats.csv
GABO_AB_DIMS=10 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/tools/ab_backward.py prof > $OUT/backward.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $OUT/backward_kernel_stats.csv
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psph -o sph -- python $R/tools/prof_sphere.py 4096 400 > /dev/null 2>&1
cp /tmp/psph/sph_kernel_stats.csv $OUT/sphere_kernel_stats.csv
# 3b. the solve launch with one and with two waves per restart (tools/ab_duo.sh)
bash $R/tools/ab_duo.sh > $OUT/ab_two_waves.txt 2>&1
cd /tmp
# 4. host timelines (no profiler)
python $R/tools/sweep_native_phases.py 64 512 > $OUT/sweep_phases.txt 2>&1
python $R/tools/plan_timeline.py > $OUT/sweep_host_timeline.txt 2>&1
ls -la $OUT
