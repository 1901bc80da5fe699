#!/bin/bash
# sweeps the refill threshold / inner steps per round / grid factor of the persistent trace kernel (default pipeline)
for cfg in "24 4 2" "16 4 2" "32 4 2" "24 3 2" "24 6 2" "24 8 2" "24 4 1" "24 4 3" "24 4 4" "16 6 3"; do
  set -- $cfg
  echo "REFILL $1 INNER $2 OVERSUB $3"
  RTAMD_REFILL=$1 RTAMD_INNER_STEPS=$2 RTAMD_TRACE_OVERSUB=$3 timeout -k 10 100 python scripts/quick_probe.py final_scene:256 random_world:128 2>/dev/null | cut -c16-110 || exit 1
done
