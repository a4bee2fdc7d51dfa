"""Batched LSD+LBD over every band schedule (12 / 40 / 200 / 500 / 700 frames in flight -> validation rounds with 32 bands / 8-band one-launch commit / 4 and 2
bands / serial kernel) on mixed texture families, each frame compared with the oracle byte for byte.
    python tools/soak_batches.py [first_seed=0] [--workers N]
--workers N: additionally the product batch driver (plf_batch_*) with N worker threads on GPU 0 (devices = [0] * N): ORB + lines +
local-map matching of 150 mixed VGA frames, 8 in flight per worker, two calls, every frame against the oracle."""
import sys, os
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd import LineSegment, matchgen
from rgbd_pl_slam_amd.synth import texture_frame

args = [a for a in sys.argv[1:] if not a.startswith("--")]
workers = 0
if "--workers" in sys.argv:
    workers = int(sys.argv[sys.argv.index("--workers") + 1])
    args = [a for a in args if a != str(workers)] if str(workers) in args[1:] else args
seed0 = int(args[0]) if args else 0
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
if workers > 0:
    from rgbd_pl_slam_amd.batch import BatchExtractor
    n = 150
    fr = np.stack(imgs[:n])
    orb_refs = list(pool.map(lambda im: orc.orb_extract(im, nfeatures=1000), imgs[:n]))
    base = next(i for i in range(n) if len(orb_refs[i]["kps"]) > 300 and len(refs[i]["kl"]) > 20)
    mp = matchgen.make_local_map(orb_refs[base]["kps"], orb_refs[base]["desc"], 3000, 5)
    ml = matchgen.make_map_lines(refs[base]["kl"], refs[base]["desc"], 400, 6)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    bounds = (0.0, 0.0, 640.0, 480.0)
    bx = BatchExtractor(nfeatures=1000, nlines=100, width=640, height=480, frames_in_flight=8, devices=[0] * workers, max_mappoints=4096, max_maplines=512)
    bx.set_local_map(mp, ml, th=3.0, nnratio=0.8, bounds=bounds)
    nbw = 0
    for rep in range(2):
        res = bx.extract(fr)
        for f in range(n):
            ro, rl = orb_refs[f], refs[f]
            ok = res[f]["kps"].tobytes() == ro["kps"].tobytes() and np.array_equal(res[f]["desc"], ro["desc"]) and \
                res[f]["lines"].tobytes() == rl["kl"].tobytes() and np.array_equal(res[f]["ldesc"], rl["desc"])
            if ok:
                rm, rn = orc.search_by_projection_map(ro["kps"], ro["desc"], None, scale, bounds, mp, 3.0, 0.8, np.full(len(ro["kps"]), -1, np.int32))
                lm, ln = orc.search_lines_by_projection(rl["kl"], rl["desc"], scale, ml, 3.0, 0.8, np.full(len(rl["kl"]), -1, np.int32))
                ok = res[f]["n_kp_matches"] == rn and np.array_equal(res[f]["match_of_kp"], rm) and res[f]["n_line_matches"] == ln and np.array_equal(res[f]["match_of_line"], lm)
            nbw += not ok
    print("batch driver, %d workers on GPU 0, %d mixed frames x 2 calls: %d mismatches" % (workers, n, nbw), flush=True)
    bad += nbw
    bx.close()
sys.exit(1 if bad else 0)
