"""NFA schedule for few frames in flight: one wave per rectangle (k_nfa_fused, with / without the table) vs k_nfa_small + staged / list kernels.
    python tools/sweep_nfa_few.py [B=1] [w=640] [h=480]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else 640
H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
sets = [np.stack([synth_frame(300 + 17 * s + i, W, H) for i in range(B)]) for s in range(4)]
ref = None
def run(**kn):
    global ref
    ls = LineSegment(nlines=100, max_width=W, max_height=H, max_batch=B)
    for k, v in kn.items():
        ls.tune(k, v)
    outs = [ls.extract_batch(im) for im in sets]
    sig = b"".join(r[0].tobytes() + r[1].tobytes() for o in outs for r in o)
    if ref is None: ref = sig
    ts = []
    for r in range(5):
        for im in sets:
            t = time.perf_counter(); ls.extract_batch(im); ts.append((time.perf_counter() - t) * 1e3)
    ls.close()
    return float(np.median(ts)), float(np.max(ts)), sig == ref
print("B=%d %dx%d: median / max ms per call over 4 frame sets" % (B, W, H))
for name, kn in (("fused, no table", dict(nfa_table=0)), ("fused + table", dict()), ("small + staged", dict(nfa_small=2)), ("small + list", dict(nfa_small=2, nfa_list=1)),
                 ("staged only", dict(nfa_fused=0, nfa_small=0, nfa_table=0))):
    m = run(**kn)
    print("  %-18s %.3f / %.3f ms  -> %.0f frames/s   same output: %s" % (name, m[0], m[1], B / m[0] * 1e3, m[2]))
