#!/bin/bash
# the default step at in-flight counts between the powers of two (the region kernel holds frames / 1024 waves per SIMD; what is left of the register file goes to the co-runners)
cd $GRAFT_REPO_ROOT
O=gpurun_out/inflight_sweep.txt; : > $O
for b in 5120 6144 7168 8192 9216; do
  echo "== batch $b" >> $O
  timeout 600 python bench.py --batch $b --no-extras --cpu-seconds 0 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
" >> $O 2>&1
done
cat $O
