import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rgbd_pl_slam_amd import ORBextractor
from rgbd_pl_slam_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h, nf = 640, 480, 1000
imgs = synth_batch(100, min(B, 16))
imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
ext = ORBextractor(nfeatures=nf, max_width=w, max_height=h, max_batch=B)
d = torch.from_numpy(imgs).cuda()
cap = ext.capacity
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ext.extract_batch_device(d, w, h, kps, desc, n, cap, s)
torch.cuda.synchronize()
t = time.time(); K = 20
for _ in range(K):
    ext.extract_batch_device(d, w, h, kps, desc, n, cap, s)
torch.cuda.synchronize()
dt = (time.time() - t) / K
print("B=%d  %.3f ms/batch  %.1f fps  n=%s" % (B, dt * 1e3, B / dt, n[:4].tolist()))
