#!/bin/bash
# the round-4 profile set, one call on the GPU box; everything lands in gpurun_out/r04_* (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python tools/pmc_traffic.py r04 > /dev/null 2>&1
cp $O/r04_pmc_traffic.json profiles/r04_pmc_traffic.json   # (bench.py reads it for roofline.traffic)
timeout 2000 python tools/pmc_sq.py --steps 2 --warmup 1 --cpu-seconds 0 --serial --no-extras > $O/r04_sq_counters.txt 2>&1
python tools/valu_counts.py $O/r04_sq_counters.txt 8192 > $O/r04_valu_counts.json
cp $O/r04_valu_counts.json profiles/r04_valu_counts.json
bash tools/timeline.sh r04 > $O/r04_timeline.txt 2>&1
bash tools/prof_r04.sh > /dev/null 2>&1
timeout 1500 python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", json.dumps(d["roofline"])[:600])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:500])
PY
