R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_i
mkdir -p $O
cd $R
python tools/pmc.py spd_tr_solve_kernel $O/pmc_solve.json -- python $R/tools/sweep_once.py > $O/pmc_solve.log 2>&1
cat $O/pmc_solve.json
