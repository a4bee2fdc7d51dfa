"""How many rectangles of a large batch enter each rect_improve stage (the staged NFA kernels of > 64 frames in flight), on the bench frame family.
    python tools/nfa_stats.py [N=2048] [distinct=256]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch_parallel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = synth_batch_parallel(0, nd, 640, 480)
imgs = np.stack([d[i % nd] for i in range(N)])
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=N)
for rep in range(2):
    res = ls.extract_batch(imgs)
c = ls.nfa_counters()
print("frames %d: rectangles entering stage 0..4: %s, not meaningful after stage 4: %d; kept lines/frame %.1f" %
      (N, list(map(int, c[:5])), int(c[5]), float(np.mean([len(r[0]) for r in res]))))
print("per frame: %s" % [round(float(x) / N, 1) for x in c[:6]])
ls.close()
