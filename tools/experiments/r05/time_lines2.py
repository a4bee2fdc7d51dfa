import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h, nl = 640, 480, 100
imgs = synth_batch(100, 16)
imgs = np.concatenate([imgs] * ((B + 15) // 16))[:B]
d = torch.from_numpy(imgs).cuda()
hs = [LineSegment(nlines=nl, max_width=w, max_height=h, max_batch=B) for _ in range(NH)]
ss = [torch.cuda.Stream(priority=-1) for _ in range(NH)]
outs = [(torch.zeros((B, nl, 17), dtype=torch.float32, device="cuda"), torch.zeros((B, nl, 32), dtype=torch.uint8, device="cuda"),
         torch.zeros((B, nl, 3), dtype=torch.float64, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(NH)]
def step(k):
    i = k % NH
    hs[i].extract_batch_device(d, w, h, outs[i][0], outs[i][1], outs[i][2], outs[i][3], nl, ss[i].cuda_stream)
for k in range(2 * NH): step(k)
torch.cuda.synchronize()
for hh in hs: hh.profile(True, True)
t = time.time(); K = 6 * NH
for k in range(K): step(k)
torch.cuda.synchronize()
dt = (time.time() - t) / K
ms = sum(hh.profile(False, True)[0] for hh in hs)
print("B=%d handles=%d  %.3f ms/batch  %.1f fps   regions avg %.1f ms" % (B, NH, dt * 1e3, B / dt, ms / K))
