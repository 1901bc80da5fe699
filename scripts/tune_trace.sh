#!/bin/bash
# sweeps the dynamic-fetch parameters of the split pipeline's trace kernel
for cfg in "24 4 2" "8 4 2" "16 4 2" "32 4 2" "48 4 2" "24 2 2" "24 8 2" "24 64 2" "24 4 1" "24 4 4" "24 4 8"; do
  set -- $cfg
  echo "REFILL $1 INNER $2 OVERSUB $3"
  RTAMD_REFILL=$1 RTAMD_INNER_STEPS=$2 RTAMD_TRACE_OVERSUB=$3 timeout -k 10 200 python scripts/perf_probe.py random_world final_scene cornell_box 2>&1 | grep split | cut -c1-21,63-140 || exit 1
done
