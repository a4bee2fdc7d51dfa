"""mean single-frame LSD+LBD latency over several synthetic frames (host in/out)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
imgs = [synth_frame(200 + i) for i in range(12)]
ls = LineSegment(nlines=100, max_width=640, max_height=480)
for im in imgs[:3]: ls.ExtractLineSegment(im)
ts = []
for im in imgs:
    t = time.perf_counter()
    for _ in range(3): ls.ExtractLineSegment(im)
    ts.append((time.perf_counter() - t) / 3 * 1e3)
print("LSD+LBD single frame over %d frames: mean %.2f ms  min %.2f  max %.2f" % (len(ts), np.mean(ts), min(ts), max(ts)))
