#!/bin/bash
# builds the stamped variants of the two f16c PDRF translation units and links libevd_vstamp.so (run on the CPU box; the .so travels)
set -e
cd "$(dirname "$0")/../../evdeblurnerf_amd"
mkdir -p lib/variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DEVD_C_STAMP"
hipcc $F -c csrc/kernel_voxel_pipe_f16c.hip -o /tmp/vs_pipe.o &
hipcc $F -mllvm -pragma-unroll-threshold=1000000 -c csrc/kernel_voxel_train_f16c.hip -o /tmp/vs_train.o &
wait
objs=$(ls lib/*.o | grep -v "/kernel_voxel_pipe_f16c.o$" | grep -v "/kernel_voxel_train_f16c.o$")
hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/vs_pipe.o /tmp/vs_train.o -o lib/variants/libevd_vstamp.so
echo lib/variants/libevd_vstamp.so
