"""One frame at a time through BOTH extractors and the four tracking matchers, as Tracking does per TrackRGBD call (Examples/RGB-D/rgbd_tum.cc:96-116 times exactly this
plus pose optimisation): ORB || LSD+LBD on two streams, then the matchers, synchronised after every frame.   python tools/latency_full.py [B=1] [config=2] [family]
Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fam = sys.argv[3] if len(sys.argv) > 3 else "polygons"
import torch
W, H, NFEAT, NLINES, _, label = bench.CONFIGS[cfg]
p = bench.Pipeline(W, H, NFEAT, NLINES, B, 0, 1234, defer_match=False, distinct=B, family=fam)
for _ in range(5): p.step()
torch.cuda.synchronize()
ts = []
for _ in range(40):
    t = time.perf_counter(); p.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
# extraction only
te = []
for _ in range(20):
    t = time.perf_counter()
    bs = p.bufs[0]
    p.lins[0].extract_batch_device(p.d_img, W, H, bs["lines"], bs["ldesc"], bs["leq"], bs["nl"], NLINES, p.sBs[0].cuda_stream)
    p.orb.extract_batch_device(p.d_img, W, H, bs["kps"], bs["desc"], bs["nk"], p.cap, p.sA.cuda_stream)
    torch.cuda.synchronize(); te.append((time.perf_counter() - t) * 1e3)
tm = []
for _ in range(20):
    t = time.perf_counter(); p.match_step(); torch.cuda.synchronize(); tm.append((time.perf_counter() - t) * 1e3)
print("%s, %d frame(s) in flight, %s: extract + match %.2f ms median (min %.2f) | both extractors %.2f ms | the four matchers alone %.2f ms | %s" %
      (label.split(":")[0], B, fam, np.median(ts), min(ts), np.median(te), np.median(tm), p.matches_frame0()))
p.close()
