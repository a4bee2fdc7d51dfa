#!/bin/bash
# the whole GPU suite, a large-batch soak, then the default bench line with its extras (few-frame figures)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python tools/soak_large.py ${1:-1000} ${2:-600} 2>&1 | tail -2
timeout 1500 python bench.py --cpu-seconds 0 --steps 8 --warmup 2 > gpurun_out/ck_bench.json 2> gpurun_out/ck_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/ck_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "region avg_launch_ms", d["roofline"]["avg_launch_ms"])
for k in ("config3_as_specified", "single_frame_latency", "pcie_inclusive"):
    print(k, json.dumps(d.get(k))[:400])
PY
