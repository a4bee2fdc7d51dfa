#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "numa or two_workers or config" 2>&1 | tail -3
timeout 1500 python bench.py --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/ck_bench.json 2> gpurun_out/ck_bench.err; tail -3 gpurun_out/ck_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/ck_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "region avg_launch_ms", d["roofline"]["avg_launch_ms"])
for k in ("fps_vs_in_flight", "config3_as_specified", "single_frame_latency", "pcie_inclusive"):
    print(k, json.dumps(d.get(k))[:700])
PY
timeout 900 python bench.py --pcie --steps 2 --warmup 1 2>&1 | tail -2 | cut -c1-1500
