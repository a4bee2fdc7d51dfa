"""per photograph: one VGA window in flight through the line extractor -- call time, region chain length, rectangles, validation rounds (fixpoint round, serial commit?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import photo_frame, natural_frame, synth_frame
import json
names = sorted(json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "tests", "golden", "real", "MANIFEST.json"))))
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
def run(tag, img):
    ts = []
    for _ in range(8):
        t = time.perf_counter(); ls.ExtractLineSegment(img); ts.append(time.perf_counter() - t)
    rs = ls.spec_rounds(1)
    print("%-22s %6.2f ms  chain %6d  rects %5d  rounds %s" % (tag, 1e3 * np.median(ts[2:]), ls.chain_lengths(1)[0], ls.rect_counts(1)[0], rs.tolist() if rs is not None else None), flush=True)
for s in range(14):
    run("photo %d %s" % (s, names[s % 7]), photo_frame(51000 + s))
for s in range(3):
    run("natural %d" % s, natural_frame(7000 + s))
    run("polygons %d" % s, synth_frame(7000 + s))
for B in (8,):
    for fam, fn, s0 in (("photo", photo_frame, 51000), ("natural", natural_frame, 41000)):
        imgs = np.stack([fn(s0 + i) for i in range(B)])
        ts = []
        for _ in range(8):
            t = time.perf_counter(); ls.extract_batch(imgs); ts.append(time.perf_counter() - t)
        print("%s x%d: %.2f ms per call, rounds %s chains %s" % (fam, B, 1e3 * np.median(ts[2:]), ls.spec_rounds(B).tolist(), ls.chain_lengths(B).tolist()), flush=True)
