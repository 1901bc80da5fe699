#!/bin/bash
# A/B of two library builds under rocprofv3 counters: scripts/pmc_ab.sh <scene> <pipe> <w> <spp> <libdir>...
# One pass per build (8 SQ counters); summaries to gpurun_out/pmc_<libdir>.txt
scene=$1; pipe=$2; w=$3; spp=$4; shift 4
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$lib${TAG:+_$TAG}
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
  rm -rf $out
  export RTAMD_LIB_DIR=$GRAFT_REPO_ROOT/raytracer-rs_amd/$lib
  (cd $GRAFT_REPO_ROOT && timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $out -o p --output-format csv -- python3 scripts/pmc_probe.py $scene $pipe $w $spp $NSYNTH > $out.log 2>&1) || { echo "rocprof failed for $lib"; tail -5 $out.log; exit 1; }
  f=$(find $out -name 'p_counter_collection.csv' | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(dirname $f) p > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
  echo "== $tag"; cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
done
