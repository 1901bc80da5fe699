#!/bin/bash
# As build_variant.sh, but the switches also reach the host side (rt_api.cpp): for constants both sides share.
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracer-rs_amd"
B=build_$name; L=lib_$name
mkdir -p $B $L
DEV="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed $*"
/opt/rocm/bin/hipcc $DEV -c csrc/rt_kernels.hip -o $B/rt_kernels.o &
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off $* -c csrc/rt_api.cpp -o $B/rt_api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $L/librtamd.so $B/rt_kernels.o build/rt_lbvh.o $B/rt_api.o build/rt_bvh.o build/rt_image.o build/rt_jpeg.o -lz
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -shared -o $L/librthost.so host/rt_host.cpp -L$L -lrtamd -Wl,-rpath,'$ORIGIN'
echo built $L
