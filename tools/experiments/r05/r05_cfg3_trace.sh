#!/bin/bash
# kernel stats of the config-3 step (8 VGA frames in flight, 2000 ORB + 200 lines, all matchers).  Run ON the GPU box.
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/c3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 20 --warmup 3 --cpu-seconds 0 --no-extras > /tmp/c3.log 2>&1
tail -1 /tmp/c3.log | cut -c1-300
python3 $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/c3 -name '*kernel_stats.csv' | head -1) 23 0.05
