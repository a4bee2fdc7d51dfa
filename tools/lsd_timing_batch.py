import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rgbd_pl_slam_amd._lib as L
L.LIB_PATH = os.environ.get("PLF_TIMING_LIB", "/tmp/plft/libplf_hip.so")
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch, natural_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w, h, nl = 640, 480, 100
imgs = (natural_batch if (len(sys.argv) > 2 and sys.argv[2] == "natural") else synth_batch)(0, 16)
imgs = np.concatenate([imgs] * ((B + 15) // 16))[:B]
d = torch.from_numpy(imgs).cuda()
ls = LineSegment(nlines=nl, max_width=w, max_height=h, max_batch=B)
o = (torch.zeros((B, nl, 17), dtype=torch.float32, device="cuda"), torch.zeros((B, nl, 32), dtype=torch.uint8, device="cuda"),
     torch.zeros((B, nl, 3), dtype=torch.float64, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"))
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ls.extract_batch_device(d, w, h, o[0], o[1], o[2], o[3], nl, s)
torch.cuda.synchronize()
L.lib().plf_lsd_timing_dump()
