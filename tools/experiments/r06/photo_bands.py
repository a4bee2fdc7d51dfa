"""hard real frames against the band count of the speculative schedule (one frame in flight): call time and fixpoint round"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
frames = [("gravel 51006", photo_frame(51006)), ("camera 51002", photo_frame(51002)), ("chelsea 51003", photo_frame(51003)), ("coffee 51004", photo_frame(51004)),
          ("astronaut 51000", photo_frame(51000)), ("brick 51001", photo_frame(51001)), ("grass 51005", photo_frame(51005)),
          ("natural 41002", natural_frame(41002)), ("natural 7000", natural_frame(7000)), ("polygons 7000", synth_frame(7000)), ("polygons 7001", synth_frame(7001))]
for NB in (8, 12, 16, 24, 32, 48, 64, 96):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    try:
        ls.tune("spec_bands", NB)
    except Exception as e:
        print("bands", NB, "rejected", e); continue
    for tag, img in frames:
        ts = []
        for _ in range(7):
            t = time.perf_counter(); ls.ExtractLineSegment(img); ts.append(time.perf_counter() - t)
        rs = ls.spec_rounds(1)
        print("bands %2d  %-16s %6.2f ms  %s" % (NB, tag, 1e3 * np.median(ts[2:]), rs.tolist() if rs is not None else None), flush=True)
    ls.close()
