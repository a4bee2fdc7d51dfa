#!/bin/bash
# second long soak of round 6 (after the real-photograph family, 64 bands / 20 rounds / 16-row reach for one frame, the wave-per-map-line line matcher): fresh seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_soak_long2.txt; : > $O
echo "== tools/soak.py 1200 32 700000 (single frames of random sizes incl. windows of real photographs + small batches, every few-frames schedule)" >> $O
timeout 1400 python tools/soak.py 1200 32 700000 2>&1 | tail -4 >> $O
for s in 71000 81000; do
  echo "== tools/soak_large.py $s 3000" >> $O
  timeout 600 python tools/soak_large.py $s 3000 2>&1 | tail -3 >> $O
done
echo "== tools/soak_batches.py 9000 --workers 4" >> $O
timeout 600 python tools/soak_batches.py 9000 --workers 4 2>&1 | tail -8 >> $O
echo "== tools/soak_match.py 500 41000 (matcher scenes: single-frame calls take the wave-per-map-line line search)" >> $O
timeout 600 python tools/soak_match.py 500 41000 2>&1 | tail -2 >> $O
grep -v amdgpu.ids $O > $O.tmp; mv $O.tmp $O; cat $O
