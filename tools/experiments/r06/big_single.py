import sys, time, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc
from rgbd_pl_slam_amd import LineSegment, ORBextractor
from rgbd_pl_slam_amd.synth import synth_frame, photo_frame, natural_frame
for fn, tag in ((synth_frame, "polygons"), (photo_frame, "photo"), (natural_frame, "natural")):
    for (w, h) in ((1280, 960), (1920, 1080)):
        im = fn(77, w, h)
        ls = LineSegment(nlines=400, max_width=w, max_height=h, max_batch=1)
        ref = orc.line_extract(im, 400)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); kl, ld, eq = ls.ExtractLineSegment(im); ts.append(time.perf_counter() - t)
        ok = kl.tobytes() == ref["kl"].tobytes() and np.array_equal(ld, ref["desc"])
        print(tag, w, h, "exact" if ok else "MISMATCH", "%.2f ms" % (1e3 * np.median(ts[1:])), (lambda r: r.tolist() if r is not None else "no validation rounds")(ls.spec_rounds(1)), flush=True)
        ls.close()
