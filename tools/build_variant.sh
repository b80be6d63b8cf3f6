#!/bin/bash
# One translation unit rebuilt with extra flags and linked against the other objects of the library:
#   bash tools/build_variant.sh <name> <file.hip> <flags...>   -> evdeblurnerf_amd/lib/variants/libevd_<name>.so   (load with EVD_LIB_PATH)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../evdeblurnerf_amd"
mkdir -p lib/variants
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c csrc/$src -o /tmp/variant_$name.o
objs=$(ls lib/*.o | grep -v "/${src%.hip}.o$")
hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/variant_$name.o -o lib/variants/libevd_$name.so
echo lib/variants/libevd_$name.so
