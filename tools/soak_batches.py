"""Batched LSD+LBD over every band schedule (12 / 40 / 200 / 500 / 700 frames in flight -> 16 / 8 / 4 / 2 bands / serial kernel) on mixed texture
families, each frame compared with the oracle byte for byte.    python tools/soak_batches.py [first_seed=0]"""
import sys, os
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import texture_frame

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = 700
pool = ThreadPoolExecutor(32)
imgs = list(pool.map(lambda s: texture_frame(seed0 + s, size=(640, 480))[0], range(N)))
refs = list(pool.map(lambda im: orc.line_extract(im, 100), imgs))
bad = 0
for B in (12, 40, 200, 500, 700):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
    res = ls.extract_batch(np.stack(imgs[:B]))
    nb = sum(not (res[f][0].tobytes() == refs[f]["kl"].tobytes() and np.array_equal(res[f][1], refs[f]["desc"])) for f in range(B))
    print("batch of %3d frames: %d mismatches" % (B, nb), flush=True)
    bad += nb
    ls.close()
sys.exit(1 if bad else 0)
