import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rgbd_pl_slam_amd._lib as L
L.LIB_PATH = os.environ.get("PLF_TIMING_LIB", "/tmp/plft/libplf_hip.so")
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
ls = LineSegment(nlines=100)
img = synth_frame(0)
for _ in range(3):
    ls.ExtractLineSegment(img)
L.lib().plf_lsd_timing_dump()
