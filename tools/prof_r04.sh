#!/bin/bash
# round-4 profile set: kernel stats of the default bench command (overlapped) and of the serial one (solo durations), copied to gpurun_out/ as r04_*
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p_def /tmp/p_ser
rocprofv3 --kernel-trace --stats -d /tmp/p_def --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_default_under_rocprofv3.json 2>/dev/null
cp $(find /tmp/p_def -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_bench_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_ser --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --serial > /dev/null 2>&1
cp $(find /tmp/p_ser -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_bench_serial_kernel_stats.csv
head -22 $GRAFT_REPO_ROOT/gpurun_out/r04_bench_serial_kernel_stats.csv | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin): print('%-26s calls %4s avg %9.3f ms' % (r['Name'].split('(')[0][:26], r['Calls'], float(r['AverageNs'])/1e6))"
