#!/bin/bash
# builds the stamped variant of the AWP per-ray kernels and links libevd_atstamp.so (run on the CPU box; the .so travels)
set -e
cd "$(dirname "$0")/../../evdeblurnerf_amd"
mkdir -p lib/variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../include -DEVD_AT_STAMP"
hipcc $F -c csrc/kernel_awp_tail.hip -o /tmp/at_stamp.o
objs=$(ls lib/*.o | grep -v "/kernel_awp_tail.o$")
hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/at_stamp.o -o lib/variants/libevd_atstamp.so
echo lib/variants/libevd_atstamp.so
