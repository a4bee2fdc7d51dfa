"""Redo rate of the speculative banded region growing (few frames in flight) per image family: polygon scenes (what its band / round thresholds were tuned on)
vs natural-image-like frames (synth.natural_frame).  Per family: frames the validation rounds could not finish (log overflow or no fixpoint within the enqueued rounds: the serial commit
wave redoes them), rounds to the fixpoint, and the LSD+LBD latency of the call.
    python tools/spec_redo.py [B=1] [frames=24]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rgbd_pl_slam_amd._lib as L
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame, texture_frame
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
fam = {"polygons": lambda s: synth_frame(300 + s), "natural": lambda s: natural_frame(300 + s), "polygons + heavy noise": lambda s: texture_frame(300 + s, kind=1, size=(640, 480))[0]}
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
print("B = %d frames in flight, 640x480, %d calls per family" % (B, N))
for name, gen in fam.items():
    fell = conv = frames = 0; rounds = []; ts = []; nl = []
    for c in range(N):
        imgs = np.stack([gen(c * B + i) for i in range(B)])
        ls.extract_batch(imgs)
        t = time.perf_counter(); res = ls.extract_batch(imgs); ts.append((time.perf_counter() - t) * 1e3)
        rs = np.zeros(4 * B, np.int32)
        if L.lib().plf_line_debug_spec_rounds(ls._h, L.vp(rs), B) == 0:
            rs = rs.reshape(B, 4); frames += B
            fell += int((rs[:, 3] != 0).sum())
            ok = rs[:, 3] == 0
            rounds += [int(r) for r in rs[ok, 2] if r > 0]
            conv += int(ok.sum())
        nl.append(len(res[0][0]))
    print("  %-24s %3d frames: %3d finished by the serial commit wave (redo rate %.1f %%), fixpoint after %.1f rounds on average (max %d); call %.2f ms median, %.2f max; "
          "lines kept %.0f" % (name, frames, fell, 100.0 * fell / max(frames, 1), float(np.mean(rounds)) if rounds else 0.0, max(rounds) if rounds else 0,
                               float(np.median(ts)), float(np.max(ts)), float(np.mean(nl))))
ls.close()
