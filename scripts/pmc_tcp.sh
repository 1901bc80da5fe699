#!/bin/bash
# L1 (TCP) behaviour of the kernels of one scene: hit rate and stalls.  scripts/pmc_tcp.sh <scene> <pipe> <w> <spp>
scene=$1; pipe=$2; w=$3; spp=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmc_tcp_$scene
rm -rf $out
(cd $ROOT && timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $out -o p --output-format csv -- python3 scripts/pmc_probe.py $scene $pipe $w $spp $NSYNTH > $out.log 2>&1) || { echo "rocprof failed"; tail -5 $out.log; exit 1; }
f=$(find $out -name 'p_counter_collection.csv' | head -1)
python3 $ROOT/scripts/pmc_summary.py $(dirname $f) p
