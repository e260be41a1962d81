# reconstruction kernel under the kernel trace (the two commands of tools/collect_profiles_r04.sh, section 4b)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prec -o rec -- python $R/tools/recon_profile.py > /tmp/recon.log 2>&1
grep nested_spd_reconstruction_kernel /tmp/prec/rec_kernel_stats.csv | sed 's/(double const.*)"//' | cut -c1-160
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/precn -o recn -- python $R/tools/recon_native_probe.py 20 > /tmp/reconn.log 2>&1
grep nested_spd_reconstruction_kernel /tmp/precn/recn_kernel_stats.csv | sed 's/(double const.*)"//' | cut -c1-160
