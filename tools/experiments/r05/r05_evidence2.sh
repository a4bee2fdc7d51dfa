#!/bin/bash
# second evidence call: the cheap round-5 files DESIGN.md cites (round log, tracking call, redo statistics, per-section timing, balance probe, region-kernel trace,
# step timeline, baseline table, short soaks).  The counter passes (pmc_traffic, pmc_sq) and the long soaks: tools/final_r05.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
bash tools/timeline.sh r05 > $O/r05_timeline.txt 2>&1
( python tools/spec_redo.py 1 24; python tools/spec_redo.py 8 12 ) > $O/r05_spec_redo.txt 2>&1
bash tools/variant_build.sh rl lsd_kernels.hip=-DPLF_ROUND_LOG line_host.hip=-DPLF_ROUND_LOG > /tmp/vb.log 2>&1
( export PLF_LIB_PATH=tools/scratch/libplf_rl.so PLF_LSD_ROUND_LOG=1; python tools/round_log.py polygons 1 3; python tools/round_log.py natural 1 3; python tools/round_log.py polygons 8 2; python tools/round_log.py natural 8 2 ) 2>&1 | grep -v amdgpu.ids > $O/r05_round_log.txt
( bash tools/r05_full_latency.sh ) > $O/r05_tracking_call.txt 2>&1
( python tools/balance_probe.py natural 8192 1024; python tools/balance_probe.py polygons 8192 1024 ) 2>&1 | grep -v amdgpu.ids > $O/r05_balance_probe.txt
bash tools/r05_regions_trace.sh > $O/r05_regions_trace.txt 2>&1
( bash tools/lsd_timing.sh && python tools/lsd_timing2.py polygons 0 && python tools/lsd_timing2.py natural 0 ) 2>&1 | grep -v amdgpu.ids > $O/r05_timing.txt
timeout 900 python tools/baseline_table.py r05 > $O/r05_baseline_table.log 2>&1
timeout 400 python tools/soak_large.py 3000 3000 > $O/r05_soak.txt 2>&1
timeout 300 python tools/soak.py 60 32 40000 >> $O/r05_soak.txt 2>&1
timeout 200 python tools/soak_match.py 40 9000 >> $O/r05_soak.txt 2>&1
tail -3 $O/r05_soak.txt; ls $O
