# runs the wave_eigh microbenchmark(s) built HERE (python tools/... hipcc line in DESIGN 7) on the GPU box: build/ubench_eigh_*
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for b in ${BINS:-build/ubench_eigh_*}; do
  echo "== $b"
  timeout 300 $b | awk '$1>=5' > gpurun_out/$(basename $b).txt
  cut -c1-8,20-64,65-150 gpurun_out/$(basename $b).txt
done
