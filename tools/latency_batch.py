"""LSD+LBD of B frames in flight through one host call: python tools/latency_batch.py [B=8] [nlines=200]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 200
imgs = synth_batch(300, B)
ls = LineSegment(nlines=NL, max_width=640, max_height=480, max_batch=B)
ls.extract_batch(imgs)
ts = []
for r in range(5):
    t = time.perf_counter(); ls.extract_batch(imgs); ts.append((time.perf_counter() - t) * 1e3)
print("B=%d LSD+LBD batch: %.2f ms (min %.2f) -> %.0f frames/s" % (B, np.mean(ts), min(ts), B / (min(ts) * 1e-3)))
