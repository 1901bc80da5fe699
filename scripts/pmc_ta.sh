#!/bin/bash
# Texture-address unit and L1 of the kernels of one scene: busy share, stalls.  scripts/pmc_ta.sh <tag> <scene> <pipe> <w> <spp>
tag=$1; scene=$2; pipe=$3; w=$4; spp=$5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmc_ta_$tag
rm -rf $out
(cd $ROOT && timeout -s KILL 150 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE -d $out -o p --output-format csv -- python3 scripts/pmc_probe.py $scene $pipe $w $spp $NSYNTH > $out.log 2>&1) || { echo "rocprof failed"; tail -5 $out.log; exit 1; }
f=$(find $out -name 'p_counter_collection.csv' | head -1)
python3 $ROOT/scripts/pmc_summary.py $(dirname $f) p
