#!/bin/bash
# exactness of the line path after a kernel change (pytest line / random / batch suites + a large-batch soak), then the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lines.py tests/test_gpu_random.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python tools/soak_large.py ${1:-0} ${2:-600} 2>&1 | tail -2
timeout 900 python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/ck_bench.json 2> gpurun_out/ck_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/ck_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "region avg_launch_ms", d["roofline"]["avg_launch_ms"])
PY
