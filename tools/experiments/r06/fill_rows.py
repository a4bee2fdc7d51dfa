"""one frame in flight: the no-growth guess of the state above a band (spec_fill rows, spec_fill_tol) instead of the warm-up growth, three image families"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
fams = {"polygons": [synth_frame(7000 + i) for i in range(12)], "natural": [natural_frame(7000 + i) for i in range(10)], "photo": [photo_frame(51000 + i) for i in range(14)]}
for fill, tol in ((0, 11.25), (4, 11.25), (8, 11.25), (16, 11.25), (8, 22.5), (16, 22.5), (32, 22.5)):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    ls.tune("spec_fill", fill); ls.tune("spec_fill_tol", tol)
    out = []
    for fam, imgs in fams.items():
        per = []
        for im in imgs:
            ts = []
            for _ in range(6):
                t = time.perf_counter(); ls.ExtractLineSegment(im); ts.append(time.perf_counter() - t)
            per.append(np.median(ts[2:]))
        out.append("%s mean %.3f ms median %.3f" % (fam, 1e3 * np.mean(per), 1e3 * np.median(per)))
    print("fill %2d tol %5.2f: %s" % (fill, tol, " | ".join(out)), flush=True)
    ls.close()
