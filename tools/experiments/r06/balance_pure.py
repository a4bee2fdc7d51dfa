"""is the region kernel of a MIXED large batch as fast as its frames allow?  8192 copies of ONE frame per launch (no imbalance possible) for windows of each photograph,
against the mixed batch of the bench (1024 distinct windows tiled to 8192): if the mixed launch lasts longer than the mean of the pure ones, the deal by defined pixels
(k_lsd_balance) leaves time on the table"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame, synth_batch_parallel
B, W, H, NL = 8192, 640, 480, 100
ls = LineSegment(nlines=NL, max_width=W, max_height=H, max_batch=B)
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
bufs = (z((B, NL, 17), torch.float32), z((B, NL, 32), torch.uint8), z((B, NL, 3), torch.float64), z(B, torch.int32))
d = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(tag):
    ls.profile(True, True)
    for _ in range(3):
        ls.extract_batch_device(d, W, H, *bufs, NL, s)
    torch.cuda.synchronize()
    ms, n = ls.profile(True, False)
    ch = ls.chain_lengths(B)
    print("%-28s region kernel %7.2f ms   chain mean %6.0f max %6d   ns per chain pixel and SIMD %.0f" % (tag, ms / n, ch.mean(), ch.max(), 1e6 * (ms / n) / (8 * ch.mean())), flush=True)
    return ms / n
pure = []
for k in range(14):
    f = torch.from_numpy(photo_frame(50000 + k)).cuda()
    d[:] = f
    pure.append(run("pure photo %d" % k))
for fam, seed in (("photo", 50000), ("natural", 40000), ("polygons", 0)):
    base = torch.from_numpy(synth_batch_parallel(seed, 1024, W, H, family=fam)).cuda()
    for lo in range(0, B, 1024):
        d[lo:lo + 1024] = base
    m = run("mixed %s (1024 distinct)" % fam)
    if fam == "photo":
        print("   mean of the 14 pure launches %.2f ms, of the 7 photographs' share of the mix" % np.mean(pure))
