#!/bin/bash
# first GPU call of round 5: where a band wave spends its cycles (polygons / natural), baseline few-frames numbers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( bash tools/lsd_timing.sh && for b in 5 20 40; do :; done
  python tools/lsd_timing2.py polygons 0; python tools/lsd_timing2.py natural 0 ) > gpurun_out/r05_timing.txt 2>&1
PLF_TIMING_EXTRA="-DPLF_LSD_TIMING_NOCNT" bash tools/lsd_timing.sh >> gpurun_out/r05_timing.txt 2>&1
( echo "== NOCNT build"; python tools/lsd_timing2.py polygons 0; python tools/lsd_timing2.py natural 0 ) >> gpurun_out/r05_timing.txt 2>&1
( python tools/spec_redo.py 1 16; python tools/spec_redo.py 8 8 ) > gpurun_out/r05_spec_redo_before.txt 2>&1
tail -50 gpurun_out/r05_timing.txt; cat gpurun_out/r05_spec_redo_before.txt
