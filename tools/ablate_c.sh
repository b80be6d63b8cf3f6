#!/bin/bash
# Ablation variants of the compensated-float16 NeRF kernel: compiles kernel_nerf_mlp_pipe_f16c.hip with -DEVD_C_ABL=<mask> (mlp_pipe_c.h)
# and links evdeblurnerf_amd/lib/abl/libevdnerf_<mask>.so from the regular objects; select one with EVD_LIB_PATH.
#   tools/ablate_c.sh 3 8 16 32 ...      (run python -m evdeblurnerf_amd.build first)
set -e
cd "$(dirname "$0")/.."
L=evdeblurnerf_amd/lib; mkdir -p $L/abl
for m in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DEVD_C_ABL=$m $EVD_C_EXTRA -c evdeblurnerf_amd/csrc/kernel_nerf_mlp_pipe_f16c.hip -o $L/abl/f16c_$m.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls $L/*.o | grep -v kernel_nerf_mlp_pipe_f16c.o) $L/abl/f16c_$m.o -o $L/abl/libevdnerf_$m.so && echo built $m ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
