"""Few-frames schedule knobs per image family: python tools/sweep_few2.py <polygons|natural> [B=1] ["k=v,k=v" ...]  (each argument = one configuration; the library is PLF_LIB_PATH)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
fam = sys.argv[1] if len(sys.argv) > 1 else "polygons"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gen = natural_frame if fam == "natural" else synth_frame
sets = [np.stack([gen(300 + 17 * s + i) for i in range(B)]) for s in range(6)]
def run(kn):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
    for k, v in kn.items(): ls.tune(k, v)
    for im in sets: ls.extract_batch(im)
    ts = []
    for r in range(3):
        for im in sets:
            t = time.perf_counter(); ls.extract_batch(im); ts.append((time.perf_counter() - t) * 1e3)
    ls.close()
    return float(np.median(ts)), float(np.mean(ts)), float(np.max(ts))
cfgs = [{}] + [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[3:]]
for kn in cfgs:
    m = run(kn)
    print("%-10s B=%d lib %-28s %-40s median %.2f mean %.2f max %.2f ms -> %.0f frames/s" % (fam, B, os.path.basename(os.environ.get("PLF_LIB_PATH", "in-tree")), kn or "default", m[0], m[1], m[2], B / m[0] * 1e3), flush=True)
