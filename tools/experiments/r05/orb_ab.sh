#!/bin/bash
# same-box A/B of orb_front.hip variants: tools/orb_ab.sh file1.hip file2.hip ...   (each replaces rgbd_pl_slam_amd/csrc/orb_front.hip in a scratch
# library; the ORB extractor alone is timed on 1024 VGA frames, 3 rounds interleaved).  Run ON the GPU box.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/rgbd_pl_slam_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fgpu-rdc -w -I$ROOT/rgbd_pl_slam_amd/csrc"
mkdir -p /tmp/oab; n=0
for v in "$@"; do /opt/rocm/bin/hipcc $FLAGS -c $ROOT/$v -o /tmp/oab/v$n.o & n=$((n+1)); done; wait
objs=$(ls *.o | grep -v orb_front.o | tr '\n' ' ')
for i in $(seq 0 $((n-1))); do /opt/rocm/bin/hipcc --offload-arch=gfx950 -fgpu-rdc --hip-link -shared -fPIC -o /tmp/oab/lib$i.so $objs /tmp/oab/v$i.o; done
for round in 1 2 3; do for i in $(seq 0 $((n-1))); do
  PLF_LIB=/tmp/oab/lib$i.so VAR=$(eval echo \${$((i+1))}) python - <<PY
import os, sys, time
sys.path.insert(0, "$ROOT")
import rgbd_pl_slam_amd._lib as L
L.LIB_PATH = os.environ["PLF_LIB"]
import numpy as np, torch
from rgbd_pl_slam_amd import ORBextractor
from rgbd_pl_slam_amd.synth import synth_frame
B = 1024
imgs = np.stack([synth_frame(i) for i in range(16)]); imgs = np.concatenate([imgs] * (B // 16))
d = torch.from_numpy(imgs).cuda()
orb = ORBextractor(nfeatures=1000, max_batch=B)
cap = orb.capacity
k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); n = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2): orb.extract_batch_device(d, 640, 480, k, ds, n, cap, s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): orb.extract_batch_device(d, 640, 480, k, ds, n, cap, s)
torch.cuda.synchronize(); print("%-40s ORB %.3f ms per 1024 frames (kp frame0 %d)" % (os.environ["VAR"], (time.perf_counter() - t0) / 8 * 1e3, int(n[0])))
PY
done; done
