#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( time python bench.py --cpu-seconds 3 ) > gpurun_out/r05_bench_try.json 2> gpurun_out/r05_bench_try.err
tail -5 gpurun_out/r05_bench_try.err
timeout 1200 python -m pytest tests/test_gpu_random.py -x -q -m gpu -k "bench_size and natural" 2>&1 | tail -3
