import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
src = open("/root/repo/tools/soak.py").read()
body = src.split("def make_image")[1].split("def eq_orb")[0]
from rgbd_pl_slam_amd.synth import synth_frame
ns = {"np": np, "synth_frame": synth_frame}
exec("def make_image" + body, ns)
import orc
from rgbd_pl_slam_amd import LineSegment
seed = int(sys.argv[1])
im, k = ns["make_image"](seed)
ref = orc.line_extract(im, 100)
print("seed", seed, "kind", k, im.shape, "oracle lines", len(ref["kl"]), flush=True)
ls = LineSegment(nlines=100, max_width=im.shape[1], max_height=im.shape[0], max_batch=8)
try:
    kl, desc, eq = ls.ExtractLineSegment(im)
    print("gpu lines", len(kl), "equal", kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"]))
    print("nseg", len(ls.segments(0)))
except Exception as ex:
    print("EXC", ex)
