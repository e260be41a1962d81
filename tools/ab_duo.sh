# development A/B (GPU box): the solve kernels of the 64- and 512-restart native sweep under rocprofv3, one wave per restart (GABO_TR_DUO=0) against two
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/duo
for n in 64 512; do
for t in 0 1; do
  GABO_TR_DUO=$t rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_${t}_$n -o s -- python $R/tools/sweep_once.py $n > /dev/null 2>&1
  cp /tmp/q_${t}_$n/s_kernel_stats.csv $R/gpurun_out/duo/stats_duo${t}_$n.csv
  echo "== GABO_TR_DUO=$t restarts=$n"; grep -i "tr_solve\|tr_start" /tmp/q_${t}_$n/s_kernel_stats.csv | cut -c1-60,200-400 | head -4
  grep -i "tr_solve" /tmp/q_${t}_$n/s_kernel_stats.csv | awk -F, '{print $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
done
done
