#!/bin/bash
# the long randomised parity soak of the shipped round-6 library, fresh seed ranges (about 35 minutes of GPU box time)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_soak_long.txt; : > $O
echo "== tools/soak.py 900 32 200000 (single frames + small batches, every few-frames schedule, ORB and lines)" >> $O
timeout 1100 python tools/soak.py 900 32 200000 2>&1 | tail -6 >> $O
for s in 31000 41000 51000; do
  echo "== tools/soak_large.py $s 3000 (3000 mixed frames per call through the large-batch kernels)" >> $O
  timeout 600 python tools/soak_large.py $s 3000 2>&1 | tail -4 >> $O
done
for s in 7000 8000; do
  echo "== tools/soak_batches.py $s --workers 2 (12-700 frames per call over every band schedule + the batch driver)" >> $O
  timeout 600 python tools/soak_batches.py $s --workers 2 2>&1 | tail -8 >> $O
done
echo "== tools/soak_match.py 300 21000 (matcher scenes)" >> $O
timeout 400 python tools/soak_match.py 300 21000 2>&1 | tail -4 >> $O
cat $O
