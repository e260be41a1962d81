cd /root/repo
for i in 1 2; do
for v in g2log r32 r64; do GABO_HIP_LIB=gabotorch_amd/libgabo_hip_$v.so python tools/ab_small_d.py $v 2>&1 | grep "d=2"; done
python tools/ab_small_d.py tab16 2>&1 | grep "d=2"
done
