#!/bin/bash
# first evidence call of the re-entered round-5 session: GPU tests, default bench line, kernel stats (overlapped / serial / natural), few-frames latency profiles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05_gputests.txt 2>&1; tail -3 $O/r05_gputests.txt
timeout 900 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err; tail -c 1500 $O/r05_bench_default.json
bash tools/prof_r05.sh > /dev/null 2>&1
bash tools/latency_profile.sh r05 > /dev/null 2>&1
bash tools/kernel_resources.sh > $O/r05_kernel_resources.txt 2>&1
ls -la $O
