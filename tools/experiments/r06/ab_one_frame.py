"""one frame in flight, LSD+LBD call over the three image families: A/B of libraries (PLF_LIB_PATH selects the build)
    PLF_LIB_PATH=tools/scratch/libplf_<tag>.so python tools/experiments/r06/ab_one_frame.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
fams = {"polygons": [synth_frame(7000 + i) for i in range(24)], "natural": [natural_frame(7000 + i) for i in range(16)], "photo": [photo_frame(51000 + i) for i in range(14)]}
for rep in range(2):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    out = []
    for fam, imgs in fams.items():
        per = []
        for im in imgs:
            ts = []
            for _ in range(7):
                t = time.perf_counter(); ls.ExtractLineSegment(im); ts.append(time.perf_counter() - t)
            per.append(np.median(ts[2:]))
        out.append("%s mean %.3f ms median %.3f" % (fam, 1e3 * np.mean(per), 1e3 * np.median(per)))
    print("%s: %s" % (os.environ.get("PLF_LIB_PATH", "shipped"), " | ".join(out)), flush=True)
    ls.close()
