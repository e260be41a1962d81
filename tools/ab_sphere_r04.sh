# round 4: variants of the sphere Gram (build here, run on the GPU box with tools/gpu_r04_b.sh)
set -e
cd /root/repo
python tools/ab_build.py sp_every2 sphere_pairwise.hip -DGABO_SPH_PW_EVERY=2 &
python tools/ab_build.py sp_every4 sphere_pairwise.hip -DGABO_SPH_PW_EVERY=4 &
python tools/ab_build.py sp_every2w3 sphere_pairwise.hip -DGABO_SPH_PW_EVERY=2 -DGABO_SPH_PW_WAVES=3 &
python tools/ab_build.py sp_every4w3 sphere_pairwise.hip -DGABO_SPH_PW_EVERY=4 -DGABO_SPH_PW_WAVES=3 &
wait
python tools/ab_build.py sp_chunks2 sphere_pairwise.hip -DGABO_SPH_CHUNKS=2 &
python tools/ab_build.py sp_chunks8 sphere_pairwise.hip -DGABO_SPH_CHUNKS=8 &
python tools/ab_build.py sp_probe1 sphere_pairwise.hip -DGABO_SPH_PROBE=1 &
python tools/ab_build.py sp_probe2 sphere_pairwise.hip -DGABO_SPH_PROBE=2 &
wait
