import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rgbd_pl_slam_amd._lib as L
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
ls = LineSegment(nlines=100)
for seed in (0, 3):
    img = synth_frame(seed)
    for _ in range(2): ls.ExtractLineSegment(img)
    st = (C.c_int32 * 8)()
    L.lib().plf_line_debug_spec_stats(ls._h, st)
    print("frame", seed, "commit %d redo %d fast_chunks %d slow_records %d | kcycles: redo %d validate %d total %d per-band set-up (after the wait for the band) %d" % tuple(st))
