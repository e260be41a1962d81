cd /root/repo
export GABO_AB_DIMS=10
for v in duow1 duow2 bwdold; do GABO_HIP_LIB=gabotorch_amd/libgabo_hip_$v.so python tools/ab_backward.py $v 2>&1 | grep "d="; done
