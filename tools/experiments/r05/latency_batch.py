"""LSD+LBD of B frames in flight through one host call: python tools/latency_batch.py [B=8] [nlines=200] [width=640] [height=480]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
H = int(sys.argv[4]) if len(sys.argv) > 4 else 480
imgs = np.stack([synth_frame(300 + i, W, H) for i in range(B)])
ls = LineSegment(nlines=NL, max_width=W, max_height=H, max_batch=B)
ls.extract_batch(imgs)
ts = []
for r in range(5):
    t = time.perf_counter(); ls.extract_batch(imgs); ts.append((time.perf_counter() - t) * 1e3)
print("B=%d %dx%d LSD+LBD batch: %.2f ms (min %.2f) -> %.0f frames/s" % (B, W, H, np.mean(ts), min(ts), B / (min(ts) * 1e-3)))
