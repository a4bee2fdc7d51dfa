#!/bin/bash
# round-5 profile set: kernel stats of the default bench command (overlapped) and of the serial one (solo durations), copied to gpurun_out/ as r05_*
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/p_def /tmp/p_ser /tmp/p_nat
rocprofv3 --kernel-trace --stats -d /tmp/p_def --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 > $O/r05_bench_default_under_rocprofv3.json 2>/dev/null
cp $(find /tmp/p_def -name "*kernel_stats.csv" | head -1) $O/r05_bench_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_ser --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --serial > /dev/null 2>&1
cp $(find /tmp/p_ser -name "*kernel_stats.csv" | head -1) $O/r05_bench_serial_kernel_stats.csv
# the natural-image family, serial: which kernel pays
rocprofv3 --kernel-trace --stats -d /tmp/p_nat --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --serial --family natural > /dev/null 2>&1
cp $(find /tmp/p_nat -name "*kernel_stats.csv" | head -1) $O/r05_bench_serial_natural_kernel_stats.csv
