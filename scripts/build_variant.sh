#!/bin/bash
# Builds librtamd.so + librthost.so with extra -D switches into raytracer-rs_amd/lib_<name> (timing experiments;
# select with RTAMD_LIB_DIR).  usage: scripts/build_variant.sh <name> <flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracer-rs_amd"
B=build_$name; L=lib_$name
mkdir -p $B $L
DEV="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed $*"
/opt/rocm/bin/hipcc $DEV -c csrc/rt_kernels.hip -o $B/rt_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $L/librtamd.so $B/rt_kernels.o build/rt_lbvh.o build/rt_api.o build/rt_bvh.o build/rt_image.o build/rt_jpeg.o -lz
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -shared -o $L/librthost.so host/rt_host.cpp -L$L -lrtamd -Wl,-rpath,'$ORIGIN'
echo built $L
