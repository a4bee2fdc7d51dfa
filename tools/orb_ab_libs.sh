#!/bin/bash
# same-box A/B of prebuilt scratch libraries (tools/variant_build.sh): tools/orb_ab_libs.sh name1 name2 ...   ("tree" = the in-tree library).  The ORB extractor
# alone is timed on 1024 VGA frames of both synthetic families, 3 rounds interleaved.  Run ON the GPU box.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2 3; do for v in "$@"; do
  if [ "$v" == "tree" ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=$ROOT/tools/scratch/libplf_$v.so; fi
  VAR=$v python - <<PY
import os, sys, time
sys.path.insert(0, "$ROOT")
import numpy as np, torch
from rgbd_pl_slam_amd import ORBextractor
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
B = 1024
out = []
for fam, gen in (("polygons", synth_frame), ("natural", natural_frame)):
    imgs = np.stack([gen(i) for i in range(16)]); imgs = np.concatenate([imgs] * (B // 16))
    d = torch.from_numpy(imgs).cuda()
    orb = ORBextractor(nfeatures=1000, max_batch=B)
    cap = orb.capacity
    k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); n = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2): orb.extract_batch_device(d, 640, 480, k, ds, n, cap, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): orb.extract_batch_device(d, 640, 480, k, ds, n, cap, s)
    torch.cuda.synchronize(); out.append("%s %.3f ms (kp0 %d)" % (fam, (time.perf_counter() - t0) / 8 * 1e3, int(n[0])))
    orb.close()
print("%-12s ORB per 1024 frames: %s" % (os.environ["VAR"], "; ".join(out)))
PY
done; done
