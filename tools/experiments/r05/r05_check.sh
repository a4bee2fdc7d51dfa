#!/bin/bash
# quick exactness + latency check of the few-frames path.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lines.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -4
for fam in polygons natural; do for B in 1 8; do python tools/latency_family.py $fam $B 8 2>&1 | grep LSD; done; done
