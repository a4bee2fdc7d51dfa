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


def test_line_keyframe_overloads_properties():
    """LSDmatcher::SearchForTriangulation / Fuse restatements (parity unpinned): structural properties of the rules"""
    rng = np.random.default_rng(5)
    d1 = rng.integers(0, 256, (80, 32), dtype=np.uint8)
    d2 = np.concatenate([d1[:50][::-1].copy(), rng.integers(0, 256, (30, 32), dtype=np.uint8)])   # 50 exact duplicates, reversed
    none = np.zeros(80, np.uint8); yes = np.ones(80, np.uint8)
    m, n = orc.lines_search_for_triangulation(d1, d2, none, none, yes, yes, 0, 0.1)
    assert n == int((m >= 0).sum()) and n >= 50
    assert np.array_equal(m[:50], np.arange(49, -1, -1))                     # the duplicate is the nearest neighbour with distance 0
    m2, n2 = orc.lines_search_for_triangulation(d1, d2, yes, none, yes, yes, 0, 0.1)
    assert n2 == 0 and (m2 == -1).all()                                      # keyframe-1 lines that hold a MapLine are never paired
    st1 = np.zeros(80, np.uint8); st1[::2] = 1
    m3, n3 = orc.lines_search_for_triangulation(d1, d2, none, none, st1, yes, 1, 0.1)
    assert (m3[1::2] == -1).all() and np.array_equal(m3[::2], m[::2])         # bOnlyStereo drops the lines without stereo data, nothing else changes
    assert orc.lines_search_for_triangulation(d1, d2[:1], none, none[:1], yes, yes[:1], 0, 0.1)[1] == 0
    best, nf = orc.lines_fuse(d2, d1, yes)
    assert np.array_equal(best[:50], np.arange(49, -1, -1)) and nf >= 50
    valid = yes.copy(); valid[:10] = 0
    best2, nf2 = orc.lines_fuse(d2, d1, valid)
    assert (best2[:10] == -1).all() and np.array_equal(best2[10:], best[10:]) and nf2 == nf - 10
    far = orc.lines_fuse(d2[50:], d1[:50], yes[:50])                         # random 256-bit descriptors are ~128 bits apart: above TH_LOW
    assert far[1] == 0
