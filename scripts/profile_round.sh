#!/bin/bash
# Profiles of one bench.py command on the GPU box, four rocprofv3 runs of the SAME command:
#   1. --kernel-trace --stats           -> per-kernel durations
#   2. --pmc FETCH_SIZE                 -> HBM read side   (own pass: FETCH_SIZE takes 3 of the 4 TCC slots)
#   3. --pmc WRITE_SIZE                 -> HBM write side
#   4. --pmc SQ_* GRBM_GUI_ACTIVE       -> VALU busy share and lanes active per VALU instruction
# usage: scripts/profile_round.sh <tag> <bench.py arguments...>      results under gpurun_out/prof_<tag>/
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 options...
  name=$1; shift
  (cd $ROOT && timeout -s KILL 400 rocprofv3 "$@" -d $OUT/$name -o p --output-format csv -- python3 bench.py "${BENCH_ARGS[@]}" > $OUT/$name.json 2> $OUT/$name.err) || { echo "$name failed"; tail -5 $OUT/$name.err; exit 1; }
  echo "$name done: $(tail -c 300 $OUT/$name.json | head -c 200)"
}
BENCH_ARGS=("$@")
run stats --kernel-trace --stats && run fetch --kernel-trace --pmc FETCH_SIZE && run write --kernel-trace --pmc WRITE_SIZE && \
run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE || exit 1
# keep only the small files: the per-dispatch traces are large
python3 $ROOT/scripts/profile_summaries.py $OUT "$tag" "python3 bench.py ${BENCH_ARGS[*]}" && find $OUT -name '*kernel_trace.csv' -delete && find $OUT -name '*counter_collection.csv' -delete
ls -la $OUT $OUT/summary
