#!/bin/bash
# Variant library with ONE source file recompiled under extra flags: tools/ablate_file.sh <name> <file.hip> <flags...>
#   -> evdeblurnerf_amd/lib/abl/libevdnerf_<name>.so  (select with EVD_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
L=evdeblurnerf_amd/lib; mkdir -p $L/abl
name=$1; src=$2; shift 2
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c evdeblurnerf_amd/csrc/$src -o $L/abl/${base}_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls $L/*.o | grep -v "/$base.o") $L/abl/${base}_$name.o -o $L/abl/libevdnerf_$name.so
echo built $name
