"""single-frame LSD+LBD latency and the speculation timeline for a few band / halo settings (the env hooks are read when a handle is created: every
configuration below makes its own handle AFTER setting them)
    python tools/spec_probe.py"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rgbd_pl_slam_amd._lib as L
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame
imgs = [synth_frame(200 + i) for i in range(6)]


def lat(ls, reps=3):
    ts = []
    for im in imgs:
        ls.ExtractLineSegment(im)
        t = time.perf_counter()
        for _ in range(reps): ls.ExtractLineSegment(im)
        ts.append((time.perf_counter() - t) / reps * 1e3)
    return np.mean(ts), min(ts), max(ts)


cfgs = [(24, 16, 0.2), (24, 16, 0.1), (24, 16, 0.05), (24, 16, 0.02), (24, 16, 0.0), (32, 16, 0.1), (32, 16, 0.05), (32, 16, 0.02), (32, 16, 0.0), (32, 12, 0.05),
        (48, 16, 0.05), (48, 16, 0.02), (48, 16, 0.0), (48, 12, 0.02), (48, 8, 0.02), (64, 16, 0.02), (64, 12, 0.02), (64, 8, 0.0)]
if len(sys.argv) > 1:
    cfgs = [tuple(float(x) if "." in x else int(x) for x in a.split(",")) for a in sys.argv[1:]]
for bands, halo, stag in cfgs:
    os.environ["PLF_LSD_SPEC_BANDS"] = str(bands); os.environ["PLF_LSD_SPEC_HALO"] = str(halo); os.environ["PLF_LSD_SPEC_STAGGER"] = str(stag)
    ls = LineSegment(nlines=100)
    m = lat(ls)
    st = (C.c_int32 * 8)()
    L.lib().plf_line_debug_spec_stats(ls._h, st)
    print("bands %2d halo %2d stagger %.2f: %.2f ms (min %.2f max %.2f) | last frame: commit %d redo %d fast %d slow %d kcyc redo %d val %d total %d setup %d" % ((bands, halo, stag) + m + tuple(st)), flush=True)
    ls.close()
os.environ["PLF_LSD_SPEC_BANDS"] = str(cfgs[-1][0]); os.environ["PLF_LSD_SPEC_HALO"] = str(cfgs[-1][1]); os.environ["PLF_LSD_SPEC_STAGGER"] = str(cfgs[-1][2]); os.environ["PLF_LSD_SPEC_TIMELINE"] = "1"
ls = LineSegment(nlines=100)
for im in imgs[:2]:
    ls.ExtractLineSegment(im); ls.ExtractLineSegment(im)
    st = (C.c_int32 * 8)()
    L.lib().plf_line_debug_spec_stats(ls._h, st)
ls.close()
