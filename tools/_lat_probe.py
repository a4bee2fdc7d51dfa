import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
imgs = [synth_frame(200 + i) for i in range(6)]
ls = LineSegment(nlines=100)
ts = []
for im in imgs:
    ls.ExtractLineSegment(im)
    t = time.perf_counter()
    for _ in range(5): ls.ExtractLineSegment(im)
    ts.append((time.perf_counter() - t) / 5 * 1e3)
print("single-frame LSD+LBD: %.2f ms (min %.2f max %.2f)" % (np.mean(ts), min(ts), max(ts)))
