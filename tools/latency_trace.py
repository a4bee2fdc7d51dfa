"""one-frame-in-flight LSD+LBD loop for rocprofv3 --kernel-trace --stats (per-kernel durations of the few-frames schedule)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
imgs = [synth_frame(200 + i) for i in range(6)]
ls = LineSegment(nlines=100, max_width=640, max_height=480)
for im in imgs[:2]: ls.ExtractLineSegment(im)
t = time.perf_counter()
for _ in range(5):
    for im in imgs: ls.ExtractLineSegment(im)
print("LSD+LBD single frame: %.2f ms" % ((time.perf_counter() - t) / 30 * 1e3))
