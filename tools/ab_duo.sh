# development A/B (GPU box): the solve kernel of the 64- and 512-restart native sweep under rocprofv3 - the product library with one wave per restart
# (GABO_TR_DUO=0) and with two, then the A/B libraries named on the command line (tools/ab_build.py): bash tools/ab_duo.sh [tag ...]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/duo
for n in 64 512; do
for t in one main "$@"; do
  L=$R/gabotorch_amd/libgabo_hip.so
  D=1
  if [ $t = one ]; then D=0; elif [ $t != main ]; then L=$R/gabotorch_amd/libgabo_hip_$t.so; fi
  GABO_TR_DUO=$D GABO_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_${t}_$n -o s -- python $R/tools/sweep_once.py $n > /dev/null 2>&1
  cp /tmp/q_${t}_$n/s_kernel_stats.csv $R/gpurun_out/duo/stats_${t}_$n.csv
  echo "$t restarts=$n: $(grep -i 'tr_solve' /tmp/q_${t}_$n/s_kernel_stats.csv | awk -F, '{print "calls", $(NF-6), "avg ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}')"
done
done
