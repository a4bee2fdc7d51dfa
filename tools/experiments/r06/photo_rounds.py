"""hard real frames (gravel, coffee, a natural-image-like one) with more validation rounds than the default 12: fixpoint round and call time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
frames = [("gravel 51006", photo_frame(51006)), ("gravel 51013", photo_frame(51013)), ("coffee 51004", photo_frame(51004)), ("camera 51002", photo_frame(51002)),
          ("chelsea 51003", photo_frame(51003)), ("grass 51005", photo_frame(51005)), ("natural 41002", natural_frame(41002)), ("polygons 7000", synth_frame(7000)), ("polygons 7001", synth_frame(7001))]
for R in (12, 16, 24, 40, 64):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    ls.tune("spec_rounds", R)
    for tag, img in frames:
        ts = []
        for _ in range(8):
            t = time.perf_counter(); ls.ExtractLineSegment(img); ts.append(time.perf_counter() - t)
        print("rounds %2d  %-16s %6.2f ms  %s" % (R, tag, 1e3 * np.median(ts[2:]), ls.spec_rounds(1).tolist()), flush=True)
    ls.close()
