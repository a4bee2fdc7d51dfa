#!/bin/bash
# solo kernel durations of the default bench step (everything on one stream) under rocprofv3; prints the top kernels.  Run ON the GPU box.
# usage: tools/prof_serial.sh <tag> [extra bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras --serial "$@" > /tmp/ps_$tag.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp $(find /tmp/ps_$tag -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${tag}_serial_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/${tag}_serial_kernel_stats.csv')))
tot=0
for r in rows:
    if r['Name'].startswith('k_'): tot+=float(r['TotalDurationNs'])/1e6/float(rows[0]['Calls'])
print("sum of solo kernel time per step: %.1f ms" % tot)
for r in rows[:24]:
    print("  %-30s calls %4s  per-step %8.3f ms  avg %8.3f ms" % (r['Name'][:30], r['Calls'], float(r['TotalDurationNs'])/1e6/float(rows[0]['Calls']), float(r['AverageNs'])/1e6))
PY
