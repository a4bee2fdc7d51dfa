#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_match.py tests/test_gpu_batch.py tests/test_gpu_cpp_mirror.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/soak_match.py 120 2>&1 | tail -4
timeout 900 python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/ck_bench.json 2> gpurun_out/ck_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/ck_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "region avg_launch_ms", d["roofline"]["avg_launch_ms"], d["matches_frame0"])
PY
