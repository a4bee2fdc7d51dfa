import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h, nl = 640, 480, 100
imgs = synth_batch(100, min(B, 16))
imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
ext = LineSegment(nlines=nl, max_width=w, max_height=h, max_batch=B)
d = torch.from_numpy(imgs).cuda()
lines = torch.zeros((B, nl, 17), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, nl, 32), dtype=torch.uint8, device="cuda")
eq = torch.zeros((B, nl, 3), dtype=torch.float64, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ext.extract_batch_device(d, w, h, lines, desc, eq, n, nl, s)
torch.cuda.synchronize()
t = time.time(); K = 5
for _ in range(K):
    ext.extract_batch_device(d, w, h, lines, desc, eq, n, nl, s)
torch.cuda.synchronize()
dt = (time.time() - t) / K
print("B=%d  %.3f ms/batch  %.1f fps  n=%s" % (B, dt * 1e3, B / dt, n[:4].tolist()))
