"""Diagnose an ORB mismatch of tools/soak.py: re-run a soak seed with its parameters and report the first stage that differs (pyramid / blur /
candidates per level).  python tools/diag_orb_params.py <seed> [<seed> ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd import ORBextractor
from rgbd_pl_slam_amd.synth import texture_frame
THREADS = 16
for seed in map(int, sys.argv[1:]):
    base = (seed // THREADS) * THREADS   # (soak.py draws the parameters per group of THREADS seeds starting at its first_seed: pass that alignment)
    first = int(os.environ.get("SOAK_FIRST", "9000"))
    gseed = first + ((seed - first) // THREADS) * THREADS
    rng = np.random.default_rng(991 + gseed)
    nf = int(rng.choice([300, 1000, 2000, 3000])); nl = int(rng.choice([50, 100, 200, 1000]))
    sf = float(np.float32(rng.choice([1.1, 1.2, 1.3, 1.5]))); nlev = int(rng.integers(3, 9)); ini = int(rng.integers(12, 40)); mn = int(rng.integers(3, 12))
    im, kind = texture_frame(seed)
    k = nlev
    while min(im.shape) / (sf ** (k - 1)) < 70: k -= 1
    h, w = im.shape
    print("seed %d kind %d size %dx%d nf %d sf %g nlev %d ini %d min %d" % (seed, kind, w, h, nf, sf, k, ini, mn))
    if im.shape == (480, 640):
        nf, sf, k, ini, mn = 1000, 1.2, 8, 20, 7
    ref = orc.orb_extract(im, nfeatures=nf, scale_factor=sf, nlevels=k, ini_th=ini, min_th=mn, debug=True)
    e = ORBextractor(nfeatures=nf, scaleFactor=sf, nlevels=k, iniThFAST=ini, minThFAST=mn, max_width=w, max_height=h)
    kps, desc = e(im)
    print("  keypoints %d vs %d, equal %s" % (len(kps), len(ref["kps"]), len(kps) == len(ref["kps"]) and kps.tobytes() == ref["kps"].tobytes() and np.array_equal(desc, ref["desc"])))
    for l in range(k):
        pyr = e.pyramid_level(0, l); bl = e.blurred_level(0, l); cand = e.candidates(0, l)
        rp = ref["pyr"][l]; rb = ref["blur"][l]
        dp = np.argwhere(pyr != rp); db = np.argwhere(bl != rb)
        print("  level %d %dx%d: pyramid diffs %d%s, blur diffs %d%s, candidates %d vs %d" % (l, pyr.shape[1] - 38, pyr.shape[0] - 38, len(dp), (" first " + str(dp[0])) if len(dp) else "",
                                                                                           len(db), (" first " + str(db[0])) if len(db) else "", len(cand), ref["ncand"][l]))
    e.close()
