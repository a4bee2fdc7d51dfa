"""8 frames in flight (BASELINE configs[2]) through ONE line handle / stream vs K handles of 8 / K frames each on K streams: what per-frame progress
of the validation rounds would buy if the library split a small batch over internal streams (every sub-batch's rounds end with ITS slowest frame, not the batch's)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch_parallel

W, H, NL, B = 640, 480, 200, 8
fam = sys.argv[1] if len(sys.argv) > 1 else "polygons"
imgs = torch.from_numpy(synth_batch_parallel(20_000, B, W, H, family=fam)).cuda()
for K in (1, 2, 4, 8):
    n = B // K
    hs = [LineSegment(nlines=NL, max_width=W, max_height=H, max_batch=n, device=0) for _ in range(K)]
    ss = [torch.cuda.Stream(priority=-1) for _ in range(K)]
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    bufs = [(z((n, NL, 17), torch.float32), z((n, NL, 32), torch.uint8), z((n, NL, 3), torch.float64), z(n, torch.int32)) for _ in range(K)]
    ts = []
    for it in range(40):
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(K):
            hs[k].extract_batch_device(imgs[k * n:(k + 1) * n], W, H, *bufs[k], NL, ss[k].cuda_stream)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ts = np.array(ts[8:]) * 1e3
    print("%s: %d handles x %d frames: median %.3f ms  min %.3f  -> %.0f frames/s" % (fam, K, n, np.median(ts), ts.min(), B / np.median(ts) * 1e3), flush=True)
    nl = torch.cat([b[3] for b in bufs]).cpu().numpy()
    print("   lines per frame", nl.tolist())
    for h_ in hs: h_.close()
