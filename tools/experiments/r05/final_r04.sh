#!/bin/bash
# the round-4 profile set, one call on the GPU box; everything lands in gpurun_out/r04_* (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python tools/pmc_traffic.py r04 > /dev/null 2>&1
cp $O/r04_pmc_traffic.json profiles/r04_pmc_traffic.json   # (bench.py reads it for roofline.traffic)
timeout 2000 python tools/pmc_sq.py --steps 2 --warmup 1 --cpu-seconds 0 --serial --no-extras > $O/r04_sq_counters.txt 2>&1
python tools/classify_isa.py $O/r04_sq_counters.txt 8192 > $O/r04_valu_classes.json
cp $O/r04_valu_classes.json profiles/r04_valu_classes.json  # (bench.py reads it for roofline.valu_issue_frac)
bash tools/timeline.sh r04 > $O/r04_timeline.txt 2>&1
bash tools/prof_r04.sh > /dev/null 2>&1
bash tools/nfa_trace.sh 8192 1024 > $O/r04_nfa_trace.txt 2>&1
bash tools/kernel_resources.sh > $O/r04_kernel_resources.txt 2>&1
( python tools/spec_redo.py 1 24; python tools/spec_redo.py 8 12 ) > $O/r04_spec_redo.txt 2>&1
( python tools/sweep_nfa_few.py 1; python tools/sweep_nfa_few.py 8 ) > $O/r04_nfa_few.txt 2>&1
timeout 1200 python tools/soak_large.py 9000 3000 > $O/r04_soak.txt 2>&1
timeout 1500 python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", json.dumps(d["roofline"])[:700])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:500])
for k in ("config3_as_specified", "single_frame_latency", "pcie_inclusive", "fps_vs_in_flight"):
    print(k, json.dumps(d.get(k))[:400])
PY
tail -2 $O/r04_soak.txt
