"""one frame in flight: 48 vs 64 bands over many frames of the three families (median LSD+LBD call)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
fams = {"polygons": [synth_frame(7000 + i) for i in range(24)], "natural": [natural_frame(7000 + i) for i in range(16)], "photo": [photo_frame(51000 + i) for i in range(14)]}
for B in (1, 2, 4):
    for NB in (40, 48, 56, 64):
        ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
        ls.tune("spec_bands", NB)
        out = []
        for fam, imgs in fams.items():
            per = []
            for k in range(0, len(imgs) - B + 1, B):
                st = np.stack(imgs[k:k + B])
                ts = []
                for _ in range(6):
                    t = time.perf_counter(); ls.extract_batch(st); ts.append(time.perf_counter() - t)
                per.append(np.median(ts[2:]))
            out.append("%s mean %.3f ms median %.3f" % (fam, 1e3 * np.mean(per), 1e3 * np.median(per)))
        print("in flight %d, bands %d: %s" % (B, NB, " | ".join(out)), flush=True)
        ls.close()
