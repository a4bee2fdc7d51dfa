"""stress the fused speculative region-growing launch: repeated single-frame and small-batch extractions must reproduce the oracle's bytes every time"""
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
imgs = [synth_frame(100 + i) for i in range(N)]
ref = [orc.line_extract(im, 100) for im in imgs]
ls = LineSegment(nlines=100, max_batch=8)
bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    for i, im in enumerate(imgs):
        kl, desc, eq = ls.ExtractLineSegment(im)
        ok = np.array_equal(desc, ref[i]["desc"]) and np.array_equal(kl["startPointX"].view(np.uint32), ref[i]["kl"]["startPointX"].view(np.uint32))
        bad += not ok
    for b0 in range(0, N - 7, 8):
        res = ls.extract_batch(np.stack(imgs[b0:b0 + 8]))
        for f in range(8):
            bad += not np.array_equal(res[f][1], ref[b0 + f]["desc"])
print("runs", "mismatches", bad)
sys.exit(1 if bad else 0)
