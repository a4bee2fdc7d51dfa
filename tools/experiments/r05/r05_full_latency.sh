#!/bin/bash
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/lf
python $GRAFT_REPO_ROOT/tools/latency_full.py 1 2 2>&1 | grep -v amdgpu.ids | tail -1
python $GRAFT_REPO_ROOT/tools/latency_full.py 1 3 2>&1 | grep -v amdgpu.ids | tail -1
python $GRAFT_REPO_ROOT/tools/latency_full.py 8 3 2>&1 | grep -v amdgpu.ids | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lf -- python $GRAFT_REPO_ROOT/tools/latency_full.py 1 2 > /tmp/lf.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/lf -name '*kernel_stats.csv' | head -1) 86 0.4 | grep -v 'spec_\|nfa_\|orb_level\|lsd_pre\|sobel\|octree\|orient\|finalize\|lbd\|rocclr\|at::'
