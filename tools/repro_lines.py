"""python tools/repro_lines.py SEED [SEED ...]: one VGA texture_frame per seed through LineSegment.ExtractLineSegment (the single-frame schedule) and, tiled to 8, through
extract_batch, against the oracle; PLF_LIB_PATH selects a scratch library.  Used to bisect the mismatch of profiles/r05_soak_long.txt."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import texture_frame
def eq_lines(got, ref):
    kl, desc, eq = got
    if len(kl) != len(ref["kl"]):
        return False
    return kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"]) and np.allclose(eq, ref["eq"], rtol=0, atol=1e-9)
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
for s in sys.argv[1:]:
    im, kind = texture_frame(int(s))
    ref = orc.line_extract(im, 100, 0)
    res = [eq_lines(ls.ExtractLineSegment(im), ref) for _ in range(4)]
    rb = ls.extract_batch(np.stack([im] * 8))
    print(os.environ.get("PLF_LIB_PATH", "in-tree"), "seed", s, "kind", kind, im.shape, "single-frame calls equal:", res, "| batch of 8 equal:", [eq_lines(r, ref) for r in rb][:3], "lines", len(ref["kl"]), flush=True)
