"""single-frame LSD+LBD latency over six synthetic frames (host in, host out: the drop-in call)
    python tools/latency_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
