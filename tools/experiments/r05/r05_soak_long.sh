#!/bin/bash
# long randomised parity soaks of the final round-5 code (large batches, every few-frames schedule incl. odd sizes, mixed batch sizes, matcher scenes).  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
( timeout 900 python tools/soak_large.py 21000 3000; timeout 900 python tools/soak_large.py 31000 3000 ) > $O/r05_soak_long.txt 2>&1
timeout 900 python tools/soak.py 700 32 60000 >> $O/r05_soak_long.txt 2>&1
timeout 600 python tools/soak_batches.py >> $O/r05_soak_long.txt 2>&1
timeout 300 python tools/soak_match.py 200 19000 >> $O/r05_soak_long.txt 2>&1
grep -v amdgpu.ids $O/r05_soak_long.txt | tail -12
