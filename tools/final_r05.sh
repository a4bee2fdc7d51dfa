#!/bin/bash
# the round-5 profile set, one call on the GPU box; everything lands in gpurun_out/r05_* (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python tools/pmc_traffic.py r05 > /dev/null 2>&1
cp $O/r05_pmc_traffic.json profiles/r05_pmc_traffic.json   # (bench.py reads it for roofline.traffic)
timeout 2000 python tools/pmc_sq.py --steps 2 --warmup 1 --cpu-seconds 0 --serial --no-extras > $O/r05_sq_counters.txt 2>&1
python tools/classify_isa.py $O/r05_sq_counters.txt 8192 > $O/r05_valu_classes.json
cp $O/r05_valu_classes.json profiles/r05_valu_classes.json  # (bench.py reads it for roofline.valu_issue_frac)
bash tools/timeline.sh r05 > $O/r05_timeline.txt 2>&1
bash tools/prof_r05.sh > /dev/null 2>&1
bash tools/kernel_resources.sh > $O/r05_kernel_resources.txt 2>&1
bash tools/latency_profile.sh r05 > /dev/null 2>&1
( python tools/spec_redo.py 1 24; python tools/spec_redo.py 8 12 ) > $O/r05_spec_redo.txt 2>&1
bash tools/variant_build.sh rl lsd_kernels.hip=-DPLF_ROUND_LOG line_host.hip=-DPLF_ROUND_LOG > /tmp/vb.log 2>&1
( export PLF_LIB_PATH=tools/scratch/libplf_rl.so PLF_LSD_ROUND_LOG=1; python tools/round_log.py polygons 1 3; python tools/round_log.py natural 1 3; python tools/round_log.py polygons 8 2; python tools/round_log.py natural 8 2 ) 2>&1 | grep -v amdgpu.ids > $O/r05_round_log.txt
( bash tools/r05_full_latency.sh ) > $O/r05_tracking_call.txt 2>&1
( python tools/balance_probe.py natural 8192 1024; python tools/balance_probe.py polygons 8192 1024 ) 2>&1 | grep -v amdgpu.ids > $O/r05_balance_probe.txt
timeout 700 python tools/soak_large.py 9000 3000 > $O/r05_soak.txt 2>&1
timeout 400 python tools/soak.py 150 32 40000 >> $O/r05_soak.txt 2>&1
timeout 300 python tools/soak_match.py 90 9000 >> $O/r05_soak.txt 2>&1
bash tools/r05_regions_trace.sh > $O/r05_regions_trace.txt 2>&1
( bash tools/lsd_timing.sh && python tools/lsd_timing2.py polygons 0 && python tools/lsd_timing2.py natural 0 ) 2>&1 | grep -v amdgpu.ids > $O/r05_timing.txt
timeout 1500 python tools/baseline_table.py r05 > $O/r05_baseline_table.log 2>&1
timeout 1500 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", json.dumps(d["roofline"])[:700])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
for k in ("config3_as_specified", "single_frame_latency", "tracking_call_latency", "fps_vs_in_flight", "natural"):
    print(k, json.dumps(d.get(k))[:900])
PY
tail -3 $O/r05_soak.txt
