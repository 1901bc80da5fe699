#!/bin/bash
# VGPRs, scratch bytes per lane and waves per SIMD of the main kernels (compiler remarks).  usage: scripts/kernel_regs.sh [extra -D flags]
cd "$(dirname "$0")/../raytracer-rs_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed -Rpass-analysis=kernel-resource-usage "$@" -c csrc/rt_kernels.hip -o /tmp/kernel_regs.o 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | paste - - - - \
  | grep -E "k_traceILi.ELb0ELb0ELi.|k_shade|k_bounceILi.ELb0ELb1ELb0|k_pathsILi.ELb0|k_finishILi.ELb0|k_intersect" \
  | sed -E 's/.*Name: ([^ ]*) .*VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Occupancy \[waves\/SIMD\]: ([0-9]+).*/\1 vgpr=\2 scratch=\3 occ=\4/' \
  | sed -E 's/_ZN12_GLOBAL__N_1[0-9]+//; s/EEvN3rtd.*( vgpr)/\1/; s/EN3rtd.*( vgpr)/\1/'
