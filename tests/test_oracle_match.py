"""Oracle matchers: sequential restatements checked for internal consistency on synthetic maps (no GPU)."""
import numpy as np

import matchgen
import orc


def _frame(seed=0):
    from rgbd_pl_slam_amd.synth import synth_frame
    r = orc.orb_extract(synth_frame(seed))
    return r["kps"], r["desc"]


def test_projection_matcher_invariants():
    kps, desc = _frame()
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    mp = matchgen.make_local_map(kps, desc, 3000, 1)
    init = np.full(len(kps), -1, np.int32)
    init[::17] = -2
    match, n = orc.search_by_projection_map(kps, desc, None, scale, (0, 0, 640, 480), mp, 3.0, 0.8, init)
    assert n > 300
    assert np.all(match[::17] == -2)                       # occupied key points are never overwritten
    got = match[match >= 0]
    assert np.all(mp["in_view"][got] == 1)
    # every accepted pair satisfies the distance gate
    for k in np.nonzero(match >= 0)[0][:200]:
        assert int(np.unpackbits(desc[k] ^ mp["desc"][match[k]]).sum()) <= 100
    # map points with Observations()>0 hold distinct key points; n counts every assignment made
    assert n >= len(got)
    # th = 1 disables the radius factor: fewer or equal matches
    _, n1 = orc.search_by_projection_map(kps, desc, None, scale, (0, 0, 640, 480), mp, 1.0, 0.8, init)
    assert n1 <= n


def test_knn2_against_bruteforce():
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (60, 32), dtype=np.uint8); t = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    t[10] = t[3]  # duplicate rows: equal distances keep the earlier index first
    idx, dist = orc.knn2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    order = np.argsort(D, axis=1, kind="stable")
    assert np.array_equal(idx, order[:, :2]) and np.array_equal(dist, np.take_along_axis(D, order[:, :2], 1))
