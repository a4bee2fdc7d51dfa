#!/bin/bash
# cumulative cost of the phases of k_orb_level: rebuilds orb_front.hip with -DOF_STOP=N (the kernel returns after phase N) into scratch libraries and
# times the ORB extractor alone on 1024 VGA frames.  Run ON the GPU box.  (Outputs are wrong for N < 6: timing only.)
cd "$(dirname "$0")/../rgbd_pl_slam_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fgpu-rdc -w"
mkdir -p /tmp/oft
for N in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc $FLAGS -DOF_STOP=$N -c orb_front.hip -o /tmp/oft/orb_front_$N.o &
done; wait
for N in 1 2 3 4 5 6; do
  objs=$(ls *.o | grep -v orb_front.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fgpu-rdc --hip-link -shared -fPIC -o /tmp/oft/libplf_$N.so $objs /tmp/oft/orb_front_$N.o
  PLF_LIB=/tmp/oft/libplf_$N.so python - <<PY
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for _ in range(5): orb.extract_batch_device(d, 640, 480, k, ds, n, cap, s)
torch.cuda.synchronize(); print("OF_STOP=$N: ORB extractor %.3f ms per 1024 frames" % ((time.perf_counter() - t0) / 5 * 1e3))
PY
done
