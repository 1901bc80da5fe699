#!/bin/bash
# HBM read bytes of every k_intersect_batch launch of scripts/c5_sort_probe.py (launch order: per bounce 4 x pixel
# order, then from bounce 1 on 4 x sorted, 4 x shuffled, then one launch with all outputs).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/c5_sort_fetch
rm -rf $out
(cd $ROOT && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o p --output-format csv -- python3 scripts/c5_sort_probe.py 10000000 3840 > $out.log 2>&1) || { echo "rocprof failed"; tail -5 $out.log; exit 1; }
f=$(find $out -name 'p_counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_intersect_batch' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
rows.sort(key=lambda r: int(r['Dispatch_Id']))
for i, r in enumerate(rows):
    print(i, r['Dispatch_Id'], "%.1f MB read (x2 corrected)" % (float(r['Counter_Value']) * 2 * 1024 / 1e6))
PY
cat $out.log
