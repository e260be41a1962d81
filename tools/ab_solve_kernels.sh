# development A/B (GPU box): the kernels of the 64-restart native sweep under rocprofv3, product library against the A/B libraries named on the command line
# (built by tools/ab_build.py): bash tools/ab_solve_kernels.sh noinl ...
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab_solve
for rep in 1 2; do
for t in main "$@"; do
  L=$R/gabotorch_amd/libgabo_hip.so
  if [ $t != main ]; then L=$R/gabotorch_amd/libgabo_hip_$t.so; fi
  GABO_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_${t}_$rep -o s -- python $R/tools/sweep_once.py 64 > /dev/null 2>&1
  cp /tmp/q_${t}_$rep/s_kernel_stats.csv $R/gpurun_out/ab_solve/stats_${t}_$rep.csv
done
done
