#!/bin/bash
# second half of the round-6 profile set, after the real-photograph family and the 20-round default went in: the figures that depend on them (the large-batch kernel
# statistics and counters of tools/final_r06.sh do not)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python tools/baseline_table.py r06 > $O/r06_baseline_table.log 2>&1
cp $O/r06_baseline_table.json profiles/r06_baseline_table.json 2>/dev/null
bash tools/latency_profile.sh r06 > /dev/null 2>&1
for f in $O/r06_latency_*; do cp $f profiles/ 2>/dev/null; done
timeout 1800 python bench.py > profiles/r06_bench_default.json 2> $O/r06_bench_default.err
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > profiles/r06_gputests.txt
mkdir -p $O/profiles_r06; cp profiles/r06_* $O/profiles_r06/
python - <<'PY'
import json
d=json.loads(open("profiles/r06_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
for k in ("config3_as_specified", "single_frame_latency", "fps_vs_in_flight", "natural", "real_photos"):
    print(k, json.dumps(d.get(k))[:900])
PY
cat profiles/r06_gputests.txt
