"""few-frames LSD+LBD loop of one image family for rocprofv3 --kernel-trace: python tools/latency_family.py <polygons|natural> [B=1] [calls=6] [nlines=100]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
fam = sys.argv[1] if len(sys.argv) > 1 else "polygons"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 6
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 100
gen = natural_frame if fam == "natural" else synth_frame
sets = [np.stack([gen(200 + c * B + i) for i in range(B)]) for c in range(N)]
ls = LineSegment(nlines=NL, max_width=640, max_height=480, max_batch=B)
for s in sets[:2]: ls.extract_batch(s)
ts = []
for s in sets:
    t = time.perf_counter(); ls.extract_batch(s); ts.append((time.perf_counter() - t) * 1e3)
print("%s B=%d: LSD+LBD call %.2f ms median (min %.2f, max %.2f) -> %.0f frames/s" % (fam, B, float(np.median(ts)), min(ts), max(ts), B / (float(np.median(ts)) * 1e-3)))
