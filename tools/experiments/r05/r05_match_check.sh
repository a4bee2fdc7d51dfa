#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_match.py tests/test_gpu_frame.py tests/test_gpu_cpp_mirror.py -x -q -m gpu 2>&1 | tail -3
bash tools/r05_full_latency.sh
