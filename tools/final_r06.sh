#!/bin/bash
# the round-6 profile set, one call on the GPU box; everything lands in gpurun_out/r06_* and the judged summaries in profiles/r06_*
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
# kernel stats: overlapped (the default command), serial (solo durations), serial on natural-image-like frames
cd /tmp; rm -rf /tmp/p_def /tmp/p_ser /tmp/p_nat
rocprofv3 --kernel-trace --stats -d /tmp/p_def --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 > $O/r06_bench_default_under_rocprofv3.json 2>/dev/null
cp $(find /tmp/p_def -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/profiles/r06_bench_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_ser --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --serial > /dev/null 2>&1
cp $(find /tmp/p_ser -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/profiles/r06_bench_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_nat --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --serial --family natural > /dev/null 2>&1
cp $(find /tmp/p_nat -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/profiles/r06_bench_serial_natural_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cp $O/r06_bench_default_under_rocprofv3.json profiles/r06_bench_default_under_rocprofv3.json
# counters: HBM traffic (overlapped command), SQ issue counters (serial command)
timeout 1500 python tools/pmc_traffic.py r06 > /dev/null 2>&1
cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
timeout 2000 python tools/pmc_sq.py --steps 2 --warmup 1 --cpu-seconds 0 --serial --no-extras > profiles/r06_sq_counters.txt 2>&1
python tools/classify_isa.py profiles/r06_sq_counters.txt 8192 > profiles/r06_valu_classes.json
python tools/per_kernel_roofline.py r06 profiles/r06_bench_serial_kernel_stats.csv 5 profiles/r06_bench_default_kernel_stats.csv 12 profiles/r06_pmc_traffic.json profiles/r06_sq_counters.txt
python tools/orb_phase_insts.py > profiles/r06_orb_phase_insts.txt 2>&1
bash tools/kernel_resources.sh > profiles/r06_kernel_resources.txt 2>&1
bash tools/timeline.sh r06 > profiles/r06_timeline.txt 2>&1
bash tools/latency_profile.sh r06 > /dev/null 2>&1
for f in $O/r06_latency_*; do cp $f profiles/ 2>/dev/null; done
# soaks of the shipped library in fresh seed ranges
timeout 700 python tools/soak_large.py 11000 3000 > profiles/r06_soak.txt 2>&1
timeout 400 python tools/soak.py 150 32 60000 >> profiles/r06_soak.txt 2>&1
timeout 300 python tools/soak_match.py 90 11000 >> profiles/r06_soak.txt 2>&1
timeout 1500 python tools/baseline_table.py r06 > $O/r06_baseline_table.log 2>&1
cp $O/r06_baseline_table.json profiles/r06_baseline_table.json 2>/dev/null
# the full default bench (with extras and the CPU baseline) and the GPU test suite on the final code
timeout 1500 python bench.py > profiles/r06_bench_default.json 2> $O/r06_bench_default.err
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > profiles/r06_gputests.txt
mkdir -p $O/profiles_r06; cp profiles/r06_* $O/profiles_r06/
python - <<'PY'
import json
d=json.loads(open("profiles/r06_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "per_kernel"})[:900])
for k in ("config3_as_specified", "single_frame_latency", "fps_vs_in_flight", "natural"):
    print(k, json.dumps(d.get(k))[:700])
PY
cat profiles/r06_gputests.txt; tail -3 profiles/r06_soak.txt
