#!/bin/bash
# frames in flight A/B of the default bench step (same box, same build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in "$@"; do
  timeout 900 python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 --batch $b > /tmp/b.json 2> /tmp/b.err
  python - $b <<'PY'
import json,sys
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("batch", sys.argv[1], "value", d["value"], "ms_per_step", d["ms_per_step"], "region avg_launch_ms", d["roofline"]["avg_launch_ms"])
PY
done
