"""one frame in flight: warm-up rows above a band (spec_halo) 0 / 1 / 2 / 4 / 8, and the reach of the warm-up regions below the band (spec_clip), three image families"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
fams = {"polygons": [synth_frame(7000 + i) for i in range(16)], "natural": [natural_frame(7000 + i) for i in range(12)], "photo": [photo_frame(51000 + i) for i in range(14)]}
for halo, clip in ((4, -1), (0, -1), (1, -1), (2, -1), (8, -1), (4, 8), (4, 16), (4, 32), (2, 8), (2, 16)):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    ls.tune("spec_halo", halo); ls.tune("spec_clip", clip)
    out = []
    for fam, imgs in fams.items():
        per = []
        for im in imgs:
            ts = []
            for _ in range(6):
                t = time.perf_counter(); ls.ExtractLineSegment(im); ts.append(time.perf_counter() - t)
            per.append(np.median(ts[2:]))
        out.append("%s mean %.3f ms median %.3f" % (fam, 1e3 * np.mean(per), 1e3 * np.median(per)))
    print("halo %d clip %3d: %s" % (halo, clip, " | ".join(out)), flush=True)
    ls.close()
