import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rgbd_pl_slam_amd import ORBextractor, LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
img = synth_frame(3)
ext = ORBextractor(nfeatures=1000, max_width=640, max_height=480)
ls = LineSegment(nlines=100, max_width=640, max_height=480)
for _ in range(3): ext(img); ls.ExtractLineSegment(img)
t = time.perf_counter()
for _ in range(20): ext(img)
print("ORB single frame (host in/out): %.2f ms" % ((time.perf_counter() - t) / 20 * 1e3))
t = time.perf_counter()
for _ in range(10): ls.ExtractLineSegment(img)
print("LSD+LBD single frame (host in/out): %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))
