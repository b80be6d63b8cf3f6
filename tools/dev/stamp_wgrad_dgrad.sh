#!/bin/bash
# builds libevd_wdstamp.so: the library with -DEVD_WD_STAMP in the f16 PDRF training unit (k_wgrad_dgrad writes its wavefronts' per-phase
# cycle sums behind the partial sets); run on the CPU box, the .so travels.  Then on the GPU box:
#   EVD_LIB_PATH=evdeblurnerf_amd/lib/variants/libevd_wdstamp.so python tools/dev/stamp_wgrad_dgrad.py
set -e
cd "$(dirname "$0")/../../evdeblurnerf_amd"
mkdir -p lib/variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DEVD_WD_STAMP"
hipcc $F -c csrc/kernel_voxel_train_f16.hip -o /tmp/wd_train.o
objs=$(ls lib/*.o | grep -v "/kernel_voxel_train_f16.o$")
hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/wd_train.o -o lib/variants/libevd_wdstamp.so
echo lib/variants/libevd_wdstamp.so
