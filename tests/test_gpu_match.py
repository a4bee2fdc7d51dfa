"""GPU parity: Hamming matchers through the C ABI vs the sequential oracle -- bit-exact (integer work)."""
import numpy as np
import pytest

import matchgen
import orc
from conftest import gpu_available

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _kp_tensor(kps):
    import torch
    return torch.from_numpy(np.frombuffer(np.ascontiguousarray(kps).tobytes(), np.uint8).copy()).cuda()


def _frame(seed=0, nf=1000):
    from rgbd_pl_slam_amd.synth import synth_frame
    r = orc.orb_extract(synth_frame(seed), nfeatures=nf)
    return r["kps"], r["desc"]


@pytest.mark.parametrize("M,th,with_ur", [(5000, 3.0, False), (2000, 1.0, True), (9000, 5.0, True)])
def test_search_by_projection_map(M, th, with_ur):
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    kps, desc = _frame()
    N = len(kps)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    rng = np.random.default_rng(M)
    mp = matchgen.make_local_map(kps, desc, M, M)
    uright = np.where(rng.uniform(0, 1, N) < 0.7, kps["x"] - rng.uniform(2, 30, N), -1).astype(np.float32) if with_ur else None
    init = np.full(N, -1, np.int32)
    init[rng.uniform(0, 1, N) < 0.1] = -2
    ref_match, ref_n = orc.search_by_projection_map(kps, desc, uright, scale, (0.0, 0.0, 640.0, 480.0), mp, th, 0.8, init)
    m = Matcher(max_keypoints=2048, max_mappoints=16384, max_batch=2)
    dk = _kp_tensor(kps); dd = _dev(desc); ds = _dev(scale)
    du = _dev(uright) if uright is not None else None
    dn = _dev(np.array([N], np.int32))
    dmp = {k: _dev(v) for k, v in mp.items()}
    match = _dev(np.stack([init, init])); nm = torch.zeros(2, dtype=torch.int32, device="cuda")
    # frame 0 passes N from host, frame 1 through the device-side counter (upper bound 2048)
    views = [Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), du), Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), du, n_device=dn)]
    m.SearchByProjection(views, dmp, th, 0.8, match, N, nm)
    torch.cuda.synchronize()
    for f in range(2):
        assert int(nm[f]) == ref_n
        assert np.array_equal(match[f].cpu().numpy(), ref_match)
    m.close()


def test_search_by_projection_map_fallback_kernel(monkeypatch):
    """candidate cache too small -> the frame is handed to the conservative fallback kernel; same result"""
    monkeypatch.setenv("PLF_MATCH_CAND_AVG", "1")
    test_search_by_projection_map(3000, 3.0, True)


def _last_frame_case(seed):
    rng = np.random.default_rng(seed)
    kps, desc = _frame(seed)
    lk, ldesc = _frame(seed + 1)
    n = len(lk)
    fx = fy = 525.0; cx, cy = 319.5, 239.5
    z = rng.uniform(0.6, 4.0, n).astype(np.float32)
    # last frame at identity; world points back-projected from last-frame key points
    xw = np.stack([(lk["x"] - cx) * z / fx, (lk["y"] - cy) * z / fy, z], 1).astype(np.float32)
    ang = 0.01
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.01, -0.005, 0.06 if seed % 2 else -0.06], np.float32)
    pose = dict(Rcw=Rcw, tcw=tcw, Rlw=np.eye(3, dtype=np.float32), tlw=np.zeros(3, np.float32), fx=fx, fy=fy, cx=cx, cy=cy, bf=40.0, b=40.0 / 525.0)
    last = dict(keys=lk, has_mappoint=(rng.uniform(0, 1, n) < 0.8).astype(np.uint8), outlier=(rng.uniform(0, 1, n) < 0.05).astype(np.uint8),
                world_pos=xw, mp_desc=matchgen.flip_bits(ldesc, rng, 20))
    # make the current frame similar to the last one so that matches exist: reuse the last descriptors at key points nearby
    return kps, desc, last, pose


@pytest.mark.parametrize("seed,mono,ori", [(0, 0, 1), (1, 0, 1), (2, 1, 0)])
def test_search_by_projection_lastframe(seed, mono, ori):
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    kps, desc, last, pose = _last_frame_case(seed)
    # current frame = last-frame key points seen from the new pose would be ideal; here both frames come from
    # independent images, so feed the LAST frame's own features as the current frame (guaranteed matches)
    kps, desc = last["keys"].copy(), matchgen.flip_bits(last["mp_desc"], np.random.default_rng(seed + 9), 10)
    N = len(kps)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    rng = np.random.default_rng(seed + 3)
    uright = np.where(rng.uniform(0, 1, N) < 0.5, kps["x"] - 40.0 / 525.0 * 525.0 / rng.uniform(0.6, 4.0, N), -1).astype(np.float32)
    init = np.full(N, -1, np.int32); init[rng.uniform(0, 1, N) < 0.05] = -2
    ref_match, ref_n = orc.search_by_projection_last(kps, desc, uright, scale, (0.0, 0.0, 640.0, 480.0), last, pose, 15.0, mono, ori, init)
    assert ref_n > 20
    m = Matcher(max_keypoints=2048)
    dk = _kp_tensor(kps); dd = _dev(desc); ds = _dev(scale); du = _dev(uright)
    cur = Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), du)
    dl = dict(keys=_kp_tensor(last["keys"]), has_mappoint=_dev(last["has_mappoint"]), outlier=_dev(last["outlier"]), world_pos=_dev(last["world_pos"]),
              mp_desc=_dev(last["mp_desc"]))
    match = _dev(init); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    m.SearchByProjectionLastFrame(cur, dl, pose, 15.0, mono, ori, match, nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == ref_n
    assert np.array_equal(match.cpu().numpy(), ref_match)
    if ori:
        # check_orientation = 2: the same search, but a key point whose assignment the rotation histogram removed reads -3 instead of -1 --
        # exactly the key points that are assigned WITHOUT the check and free WITH it (the plf.hpp adapter sets those to NULL as the reference does)
        m2 = _dev(init); m0 = _dev(init); n2 = torch.zeros(1, dtype=torch.int32, device="cuda"); n0 = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchByProjectionLastFrame(cur, dl, pose, 15.0, mono, 2, m2, n2)
        m.SearchByProjectionLastFrame(cur, dl, pose, 15.0, mono, 0, m0, n0)
        torch.cuda.synchronize()
        a2, a0 = m2.cpu().numpy(), m0.cpu().numpy()
        assert int(n2[0]) == ref_n and np.array_equal(np.where(a2 == -3, -1, a2), ref_match)
        culled = a2 == -3
        assert culled.sum() == int(n0[0]) - ref_n or culled.sum() > 0 or int(n0[0]) == ref_n     # (overwritten key points can be culled more than once per key point)
        assert np.all(a0[culled] >= 0) and np.all(ref_match[culled] == -1)
    m.close()


def test_line_matchers():
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd.synth import synth_frame
    a = orc.line_extract(synth_frame(0), 100); b = orc.line_extract(synth_frame(1), 100)
    rng = np.random.default_rng(0)
    cur_desc = np.concatenate([matchgen.flip_bits(a["desc"][:70], rng, 25), b["desc"][:30]])
    m = Matcher(max_lines=512, max_mappoints=2048)
    # BF kNN (k = 2), cv::batchDistance tie order
    idx, dist = orc.knn2(a["desc"], cur_desc)
    dm = m.knnMatch(_dev(a["desc"]), _dev(cur_desc))
    assert np.array_equal(dm["trainIdx"], idx) and np.array_equal(dm["distance"], dist.astype(np.float32))
    assert np.array_equal(dm["queryIdx"], np.repeat(np.arange(len(idx)), 2).reshape(-1, 2))
    # last-frame line tracking with MAD threshold
    has_ml = (rng.uniform(0, 1, len(a["desc"])) < 0.8).astype(np.uint8)
    ref_match, ref_n = orc.match_lines_knn(a["desc"], cur_desc, has_ml)
    match = torch.full((len(cur_desc),), -1, dtype=torch.int32, device="cuda"); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_last, d_cur, d_has = _dev(a["desc"]), _dev(cur_desc), _dev(has_ml)
    m.SearchLinesLastFrame(d_last, d_cur, d_has, match, nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == ref_n and ref_n > 10
    assert np.array_equal(match.cpu().numpy(), ref_match)
    # projection search against map lines
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    ml = matchgen.make_map_lines(a["kl"], a["desc"], 500, 3)
    init = np.full(len(a["kl"]), -1, np.int32); init[::11] = -2
    ref_match, ref_n = orc.search_lines_by_projection(a["kl"], a["desc"], scale, ml, 3.0, 0.8, init)
    assert ref_n > 5
    dkl = torch.from_numpy(np.frombuffer(np.ascontiguousarray(a["kl"]).tobytes(), np.uint8).copy()).cuda()
    dld = _dev(a["desc"]); dsc = _dev(scale)   # views hold raw addresses: keep the tensors alive
    view = Matcher.lineframe_view(len(a["kl"]), dkl, dld, dsc)
    dml = {k: _dev(v) for k, v in ml.items()}
    match = _dev(init); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    m.SearchLinesByProjection([view], dml, 3.0, 0.8, match, len(a["kl"]), nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == ref_n
    assert np.array_equal(match.cpu().numpy(), ref_match)
    m.close()


def test_lastframe_matchers_batched():
    """plf_match_project_lastframe_batch / plf_match_lines_lastframe_batch: several independent current frames against ONE last frame, every
    frame with its own pose, key point count (device-resident) and line set -- each frame must equal the single-frame oracle result"""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd.synth import synth_frame
    B, stride = 5, 1100
    _, _, last, pose0 = _last_frame_case(4)
    # half of the last frame's map points have no observations (localisation mode): overwritable assignments inside the batch too
    last["obs_positive"] = (np.random.default_rng(5).uniform(0, 1, len(last["keys"])) < 0.5).astype(np.uint8)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    bounds = (0.0, 0.0, 640.0, 480.0)
    kps_all = np.zeros((B, stride), last["keys"].dtype); desc_all = np.zeros((B, stride, 32), np.uint8); ur_all = np.full((B, stride), -1, np.float32)
    n_all = np.zeros(B, np.int32); init_all = np.full((B, stride), -1, np.int32)
    poses, refs = [], []
    for f in range(B):
        rng = np.random.default_rng(50 + f)
        n = int(len(last["keys"]) * (1.0 - 0.1 * f))
        k = last["keys"][:n].copy(); k["x"] += rng.normal(0, 1.5, n).astype(np.float32); k["y"] += rng.normal(0, 1.5, n).astype(np.float32)
        d = matchgen.flip_bits(last["mp_desc"][:n], rng, 5 + 10 * f)
        ur = np.where(rng.uniform(0, 1, n) < 0.5, k["x"] - 40.0 / rng.uniform(0.6, 4.0, n), -1).astype(np.float32)
        init = np.full(n, -1, np.int32); init[rng.uniform(0, 1, n) < 0.05] = -2
        pose = dict(pose0); pose["tcw"] = (pose0["tcw"] * (1.0 + 0.3 * f) * (-1 if f == 2 else 1)).astype(np.float32)
        rm, rn = orc.search_by_projection_last(k, d, ur, scale, bounds, last, pose, 15.0, 0, 1, init)
        kps_all[f, :n] = k; desc_all[f, :n] = d; ur_all[f, :n] = ur; n_all[f] = n; init_all[f, :n] = init
        poses.append(pose); refs.append((rm, rn, n))
    assert sum(r[1] for r in refs) > 300
    m = Matcher(max_keypoints=stride, max_mappoints=64, max_lines=512, max_batch=B)
    dk = torch.from_numpy(np.frombuffer(kps_all.tobytes(), np.uint8).copy()).cuda(); dd = _dev(desc_all); du = _dev(ur_all); dn = _dev(n_all); ds = _dev(scale)
    views = [Matcher.frame_view(stride, dk.data_ptr() + f * stride * 28, dd.data_ptr() + f * stride * 32, ds, bounds, du.data_ptr() + f * stride * 4, dn.data_ptr() + 4 * f)
             for f in range(B)]
    dl = dict(keys=_kp_tensor(last["keys"]), has_mappoint=_dev(last["has_mappoint"]), outlier=_dev(last["outlier"]), world_pos=_dev(last["world_pos"]),
              mp_desc=_dev(last["mp_desc"]), obs_positive=_dev(last["obs_positive"]))
    match = _dev(init_all); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    for rep in range(2):   # second call: cached frame / pose tables
        match.copy_(torch.from_numpy(init_all))
        m.SearchByProjectionLastFrameBatch(views, dl, poses, 15.0, 0, 1, match, stride, nm)
        torch.cuda.synchronize()
        got = match.cpu().numpy()
        for f, (rm, rn, n) in enumerate(refs):
            assert int(nm[f]) == rn, "frame %d" % f
            assert np.array_equal(got[f, :n], rm), "frame %d" % f
    # lines: one last frame's LBD descriptors against B current line sets of different sizes (one with a single line: no 2-NN, zero matches)
    a = orc.line_extract(synth_frame(20), 150)
    lstride = 160
    ld_all = np.zeros((B, lstride, 32), np.uint8); ln_all = np.zeros(B, np.int32); lrefs = []
    has_ml = (np.random.default_rng(8).uniform(0, 1, len(a["desc"])) < 0.8).astype(np.uint8)
    for f in range(B):
        rng = np.random.default_rng(80 + f)
        b = orc.line_extract(synth_frame(21 + f), 150)
        cur = np.concatenate([matchgen.flip_bits(a["desc"][:100 - 20 * f], rng, 10 * f), b["desc"][:40]])
        if f == 3:
            cur = cur[:1]
        cur = np.ascontiguousarray(cur[rng.permutation(len(cur))])
        ld_all[f, :len(cur)] = cur; ln_all[f] = len(cur)
        if len(cur) >= 2:
            lrefs.append(orc.match_lines_knn(a["desc"], cur, has_ml) + (len(cur),))
        else:
            lrefs.append((np.full(len(cur), -1, np.int32), 0, len(cur)))
    dld = _dev(ld_all); dln = _dev(ln_all); d_last = _dev(a["desc"]); d_has = _dev(has_ml)
    lviews = [Matcher.lineframe_view(lstride, 0, dld.data_ptr() + f * lstride * 32, ds, dln.data_ptr() + 4 * f) for f in range(B)]
    lmatch = torch.full((B, lstride), -1, dtype=torch.int32, device="cuda"); lnm = torch.zeros(B, dtype=torch.int32, device="cuda")
    m.SearchLinesLastFrameBatch(d_last, d_has, lviews, lmatch, lstride, lnm)
    torch.cuda.synchronize()
    got = lmatch.cpu().numpy()
    for f, (rm, rn, n) in enumerate(lrefs):
        assert int(lnm[f]) == rn, "line frame %d" % f
        assert np.array_equal(got[f, :n], rm), "line frame %d" % f
    assert sum(r[1] for r in lrefs) > 100
    m.close()


LINE_SCENES = [
    # frames a / b, lines kept, how many of a's lines reappear, bit flips, share of last-frame lines with a MapLine, map lines, th, nnratio
    dict(sa=0, sb=1, n=100, keep=70, flips=25, has=0.8, M=500, th=3.0, nn=0.8),
    dict(sa=2, sb=3, n=200, keep=150, flips=10, has=1.0, M=800, th=1.0, nn=0.9),    # th == 1: radius not scaled; every line tracked
    dict(sa=4, sb=5, n=400, keep=100, flips=60, has=0.5, M=2000, th=5.0, nn=0.6),   # heavy noise: MAD threshold bites, more map lines than lines
    dict(sa=6, sb=7, n=50, keep=50, flips=0, has=0.9, M=64, th=3.0, nn=0.8),        # exact copies: distance 0, ties in the kNN
    dict(sa=8, sb=9, n=30, keep=3, flips=5, has=0.3, M=10, th=3.0, nn=0.8),         # few lines, tiny map
    dict(sa=10, sb=11, n=100, keep=0, flips=0, has=0.8, M=300, th=3.0, nn=0.8),     # nothing in common
    dict(sa=12, sb=13, n=120, keep=120, flips=3, has=0.0, M=300, th=3.0, nn=0.8),   # no last-frame line holds a MapLine: zero matches
]


@pytest.mark.parametrize("sc", LINE_SCENES, ids=lambda sc: "a%d_n%d_keep%d_flips%d" % (sc["sa"], sc["n"], sc["keep"], sc["flips"]))
def test_line_matcher_scenes(sc):
    """rows 13 / 14 (LSDmatcher::SearchByProjection x3, include/LSDmatcher.h:32,35,40): BF kNN + MAD rule and the projection search on scenes
    that differ in size, noise, occupancy and thresholds"""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd.synth import synth_frame
    a = orc.line_extract(synth_frame(sc["sa"]), sc["n"]); b = orc.line_extract(synth_frame(sc["sb"]), sc["n"])
    rng = np.random.default_rng(100 + sc["sa"])
    keep = min(sc["keep"], len(a["desc"]))
    cur_desc = np.concatenate([matchgen.flip_bits(a["desc"][:keep], rng, sc["flips"]), b["desc"][:max(len(b["desc"]) - keep, 2)]])
    cur_desc = np.ascontiguousarray(cur_desc[rng.permutation(len(cur_desc))])
    m = Matcher(max_lines=1024, max_mappoints=4096)
    idx, dist = orc.knn2(a["desc"], cur_desc)
    dm = m.knnMatch(_dev(a["desc"]), _dev(cur_desc))
    assert np.array_equal(dm["trainIdx"], idx) and np.array_equal(dm["distance"], dist.astype(np.float32))
    has_ml = (rng.uniform(0, 1, len(a["desc"])) < sc["has"]).astype(np.uint8)
    ref_match, ref_n = orc.match_lines_knn(a["desc"], cur_desc, has_ml)
    match = torch.full((len(cur_desc),), -1, dtype=torch.int32, device="cuda"); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_last, d_cur, d_has = _dev(a["desc"]), _dev(cur_desc), _dev(has_ml)
    m.SearchLinesLastFrame(d_last, d_cur, d_has, match, nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == ref_n and np.array_equal(match.cpu().numpy(), ref_match)
    if sc["has"] == 0.0:
        assert ref_n == 0
    # the keyframe overload (LSDmatcher.h:35) is the same rule with the keyframe's descriptors on the query side: swap the roles
    has_kf = (rng.uniform(0, 1, len(cur_desc)) < 0.7).astype(np.uint8)
    ref_match2, ref_n2 = orc.match_lines_knn(cur_desc, a["desc"], has_kf)
    match2 = torch.full((len(a["desc"]),), -1, dtype=torch.int32, device="cuda")
    d_has2 = _dev(has_kf)
    m.SearchLinesLastFrame(d_cur, d_last, d_has2, match2, nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == ref_n2 and np.array_equal(match2.cpu().numpy(), ref_match2)
    # projection search against map lines
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    ml = matchgen.make_map_lines(a["kl"], a["desc"], sc["M"], 3 + sc["sa"])
    init = np.full(len(a["kl"]), -1, np.int32); init[::7] = -2
    ref_match, ref_n = orc.search_lines_by_projection(a["kl"], a["desc"], scale, ml, sc["th"], sc["nn"], init)
    dkl = torch.from_numpy(np.frombuffer(np.ascontiguousarray(a["kl"]).tobytes(), np.uint8).copy()).cuda()
    dld = _dev(a["desc"]); dsc = _dev(scale)
    view = Matcher.lineframe_view(len(a["kl"]), dkl, dld, dsc)
    dml = {k: _dev(v) for k, v in ml.items()}
    match = _dev(init); nm2 = torch.zeros(1, dtype=torch.int32, device="cuda")
    m.SearchLinesByProjection([view], dml, sc["th"], sc["nn"], match, len(a["kl"]), nm2)
    torch.cuda.synchronize()
    assert int(nm2[0]) == ref_n and np.array_equal(match.cpu().numpy(), ref_match)
    m.close()


@pytest.mark.parametrize("force", [True, False], ids=["forced_on_a_small_frame", "more_lines_than_fit_the_lds"])
def test_line_projection_search_global_memory_variant(monkeypatch, force):
    """k_match_project_lines stages 56 bytes per line in LDS, which fits up to 2688 lines while plf_matcher_create accepts max_lines up to 18000:
    frames above that run the variant that reads the lines from global memory (8 bytes of LDS per line).  Same bits as the oracle, on a small frame with the
    variant forced (PLF_MATCH_LINES_STAGE_MAX=0) and on a frame of > 2688 lines (the lines of 13 synthetic frames together)."""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd.synth import synth_frame
    if force:
        monkeypatch.setenv("PLF_MATCH_LINES_STAGE_MAX", "0")
        a = orc.line_extract(synth_frame(0), 100)
        kl, desc, M = a["kl"], a["desc"], 500
    else:
        parts = [orc.line_extract(synth_frame(60 + i), 400) for i in range(13)]   # (~250 lines each)
        kl = np.concatenate([p_["kl"] for p_ in parts]); desc = np.concatenate([p_["desc"] for p_ in parts])
        assert len(kl) > 2688
        M = 1500
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    ml = matchgen.make_map_lines(kl, desc, M, 11)
    init = np.full(len(kl), -1, np.int32); init[::9] = -2
    ref_match, ref_n = orc.search_lines_by_projection(kl, desc, scale, ml, 3.0, 0.8, init)
    assert ref_n > 10
    m = Matcher(max_lines=4096, max_mappoints=4096)
    dkl = torch.from_numpy(np.frombuffer(np.ascontiguousarray(kl).tobytes(), np.uint8).copy()).cuda()
    dld = _dev(desc); dsc = _dev(scale)
    view = Matcher.lineframe_view(len(kl), dkl, dld, dsc)
    dml = {k: _dev(v) for k, v in ml.items()}
    match = _dev(init); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    m.SearchLinesByProjection([view], dml, 3.0, 0.8, match, len(kl), nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == ref_n and np.array_equal(match.cpu().numpy(), ref_match)
    m.close()


@pytest.mark.parametrize("wave_max", ["64", "0"], ids=["wave_per_map_line", "thread_per_map_line"])
def test_line_projection_search_wave_and_thread_kernels(monkeypatch, wave_max):
    """Round 6: up to 64 frames per call the line projection search runs one wave per map line with the key lines across the lanes (k_match_project_lines_w); larger
    batches keep the thread-per-map-line kernel.  Both on the same scenes (PLF_MATCH_LINES_WAVE_MAX is read per call): crowded windows (1500 map lines for ~250
    lines: several rounds), descriptors duplicated so that best and second-best tie at distance 0 and at equal distances (the reference's first-come order), pre-set
    matches, th = 1 and th != 1 -- and a call of 70 frames (always the thread kernel) next to one of 64."""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd.synth import synth_frame, photo_frame
    monkeypatch.setenv("PLF_MATCH_LINES_WAVE_MAX", wave_max)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    m = Matcher(max_lines=1024, max_mappoints=4096, max_batch=70)
    rng = np.random.default_rng(8)
    views, refs, keep = [], [], []
    for k, (img, nl, M, th, nn) in enumerate([(synth_frame(70), 400, 1500, 3.0, 0.8), (photo_frame(9), 300, 1200, 1.0, 0.9), (synth_frame(71), 100, 500, 3.0, 0.8),
                                             (photo_frame(11), 64, 300, 2.0, 0.7), (synth_frame(72), 65, 700, 3.0, 1.0)]):
        a = orc.line_extract(img, nl)
        kl, desc = a["kl"].copy(), a["desc"].copy()
        n = len(kl)
        assert n >= 40
        # ties: every third line gets the descriptor of its predecessor (equal distances to every map line; when they share the octave the ratio test sees them)
        desc[2::3] = desc[1::3][:len(desc[2::3])]
        ml = matchgen.make_map_lines(kl, desc, M, 20 + k)
        init = np.full(n, -1, np.int32); init[::7] = -2; init[3::11] = 5
        ref_match, ref_n = orc.search_lines_by_projection(kl, desc, scale, ml, th, nn, init)
        dkl = torch.from_numpy(np.frombuffer(np.ascontiguousarray(kl).tobytes(), np.uint8).copy()).cuda()
        dld = _dev(desc); dsc = _dev(scale)
        view = Matcher.lineframe_view(n, dkl, dld, dsc)
        dml = {kk: _dev(v) for kk, v in ml.items()}
        match = _dev(init); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchLinesByProjection([view], dml, th, nn, match, n, nm)
        torch.cuda.synchronize()
        assert int(nm[0]) == ref_n and np.array_equal(match.cpu().numpy(), ref_match), k
        if k == 2:
            keep = (view, dml, init, ref_match, ref_n, n, (dkl, dld, dsc))
    assert keep[4] > 10
    view, dml, init, ref_match, ref_n, n, _alive = keep
    for B in (64, 70):      # 64: the wave kernel (unless forced off); 70: always the thread kernel
        match = _dev(np.tile(init, (B, 1))); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
        m.SearchLinesByProjection([view] * B, dml, 3.0, 0.8, match, n, nm)
        torch.cuda.synchronize()
        assert (nm.cpu().numpy() == ref_n).all() and (match.cpu().numpy() == ref_match[None]).all(), B
    m.close()


LINE_KF_SCENES = [
    # lines in keyframe 1 / 2, how many of keyframe 1's lines reappear in keyframe 2, bit flips, share with a MapLine (kf1, kf2), share with stereo, bOnlyStereo, MAD factor
    dict(s1=30, s2=31, n=120, keep=90, flips=12, ml1=0.3, ml2=0.3, st=0.7, only=0, f=0.1),
    dict(s1=32, s2=33, n=200, keep=60, flips=40, ml1=0.0, ml2=0.0, st=0.5, only=1, f=0.1),
    dict(s1=34, s2=35, n=60, keep=60, flips=0, ml1=0.5, ml2=0.1, st=1.0, only=1, f=0.5),     # exact copies: distance 0 and ties
    dict(s1=36, s2=37, n=40, keep=1, flips=3, ml1=1.0, ml2=0.0, st=0.0, only=0, f=0.1),      # every keyframe-1 line already has a MapLine: nothing to pair
    dict(s1=38, s2=39, n=150, keep=100, flips=25, ml1=0.2, ml2=0.6, st=0.0, only=1, f=0.0),  # no stereo data at all with bOnlyStereo: nothing; factor 0
]


@pytest.mark.parametrize("sc", LINE_KF_SCENES, ids=lambda sc: "kf%d_n%d_keep%d_only%d" % (sc["s1"], sc["n"], sc["keep"], sc["only"]))
def test_line_triangulation_and_fuse(sc):
    """SURVEY 8f rank 3, the two remaining LSDmatcher overloads (include/LSDmatcher.h:54,58): SearchForTriangulation (kNN + MAD, unmatched lines only,
    bOnlyStereo) and Fuse (nearest keyframe line over all of them, TH_LOW) -- bit-equal to the oracle; "parity unpinned" like rows 13 / 14"""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd.synth import synth_frame
    a = orc.line_extract(synth_frame(sc["s1"]), sc["n"]); b = orc.line_extract(synth_frame(sc["s2"]), sc["n"])
    rng = np.random.default_rng(200 + sc["s1"])
    keep = min(sc["keep"], len(a["desc"]))
    d1 = a["desc"]
    d2 = np.concatenate([matchgen.flip_bits(d1[:keep], rng, sc["flips"]), b["desc"][:max(len(b["desc"]) - keep, 2)]])
    d2 = np.ascontiguousarray(d2[rng.permutation(len(d2))])
    ml1 = (rng.uniform(0, 1, len(d1)) < sc["ml1"]).astype(np.uint8); ml2 = (rng.uniform(0, 1, len(d2)) < sc["ml2"]).astype(np.uint8)
    st1 = (rng.uniform(0, 1, len(d1)) < sc["st"]).astype(np.uint8); st2 = (rng.uniform(0, 1, len(d2)) < sc["st"]).astype(np.uint8)
    ref, rn = orc.lines_search_for_triangulation(d1, d2, ml1, ml2, st1, st2, sc["only"], sc["f"])
    m = Matcher(max_lines=1024, max_mappoints=64)
    match = torch.zeros(len(d1), dtype=torch.int32, device="cuda"); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    t = [_dev(x) for x in (d1, d2, ml1, ml2, st1, st2)]
    m.SearchLinesForTriangulation(t[0], t[1], t[2], t[3], t[4], t[5], sc["only"], match, nm, mad_factor=sc["f"])
    torch.cuda.synchronize()
    assert int(nm[0]) == rn and np.array_equal(match.cpu().numpy(), ref)
    if sc["ml1"] == 1.0 or (sc["only"] and sc["st"] == 0.0):
        assert rn == 0
    # fewer than two keyframe-2 lines: knnMatch(k = 2) has no second neighbour
    m.SearchLinesForTriangulation(t[0], t[1][:1], t[2], t[3][:1], t[4], t[5][:1], 0, match, nm)
    torch.cuda.synchronize()
    assert int(nm[0]) == 0 and (match.cpu().numpy() == -1).all()
    # Fuse: the keyframe's lines = d2, map lines = descriptors of keyframe 1's lines with a few flips (true duplicates) + unrelated ones
    mld = np.concatenate([matchgen.flip_bits(d1[:keep], rng, 2 * sc["flips"]), rng.integers(0, 256, (25, 32), dtype=np.uint8)])
    valid = (rng.uniform(0, 1, len(mld)) < 0.85).astype(np.uint8)
    fb, fn = orc.lines_fuse(d2, mld, valid)
    best = torch.zeros(len(mld), dtype=torch.int32, device="cuda"); nf = torch.zeros(1, dtype=torch.int32, device="cuda")
    dk, dm, dv = _dev(d2), _dev(mld), _dev(valid)
    m.FuseLines(dk, dm, dv, best, nf)
    torch.cuda.synchronize()
    assert int(nf[0]) == fn and np.array_equal(best.cpu().numpy(), fb)
    if keep > 10 and sc["flips"] <= 12:
        assert fn > 5
    m.close()


def test_descriptor_distance_and_matrix():
    _need_gpu()
    import ctypes as C
    from rgbd_pl_slam_amd import DescriptorDistance, _lib as L
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (50, 32), dtype=np.uint8); b = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    D = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2).astype(np.int32)
    assert DescriptorDistance(a[3], b[5]) == D[3, 5]
    out = np.zeros((50, 70), np.int32)
    L.check(L.lib().plf_hamming256_matrix(L.vp(a), 50, L.vp(b), 70, L.vp(out), L.MEM_HOST, 0, None), "plf_hamming256_matrix")
    assert np.array_equal(out, D)


def test_gpu_equals_reference_search_by_projection_fixture():
    """HIP matcher vs tests/golden/ref_glue_search_map.json: the result of the reference binary's own
    ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) on the same inputs."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_search_map_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_search_map.json")):
        N = len(c["kps"])
        m = Matcher(max_keypoints=1024, max_mappoints=4096, max_batch=1)
        dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"])
        du = _dev(c["uright"]) if c["uright"] is not None else None
        dmp = {k: _dev(v) for k, v in c["mp"].items()}
        match = _dev(c["init"][None, :]); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchByProjection([Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), du)], dmp, c["th"], 0.8, match, N, nm)
        torch.cuda.synchronize()
        assert int(nm[0]) == c["nmatches"]
        assert np.array_equal(match[0].cpu().numpy(), c["match"])
        m.close()


def _bow_random_case(seed, nkf, nf, nnodes, shared=False):
    """random (keyframe, frame) pair in the flattened layout; shared=True lists some frame features in two nodes"""
    rng = np.random.default_rng(seed)
    kf_desc = rng.integers(0, 256, (nkf, 32), dtype=np.uint8)
    src = rng.integers(0, nkf, nf)
    f_desc = matchgen.flip_bits(kf_desc[src].copy(), rng, 35)
    rnd = rng.uniform(0, 1, nf) < 0.3
    f_desc[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
    knode = rng.integers(0, nnodes, nkf).astype(np.uint32) * 5 + 100
    fnode = np.where(rng.uniform(0, 1, nf) < 0.85, knode[src], rng.integers(0, nnodes, nf).astype(np.uint32) * 5 + 100 + rng.integers(0, 2, nf).astype(np.uint32) * 2).astype(np.uint32)
    kf_angle = rng.uniform(0, 360, nkf).astype(np.float32)
    f_angle = np.mod(kf_angle[src] - np.where(rng.uniform(0, 1, nf) < 0.8, rng.uniform(20, 28, nf), rng.uniform(0, 360, nf)), 360).astype(np.float32)
    has = (rng.uniform(0, 1, nkf) < 0.8).astype(np.uint8)

    def flat(node, extra=None):
        ids = np.unique(node)
        lists = [list(np.nonzero(node == i)[0]) for i in ids]
        if extra is not None:
            for j, k in extra:
                lists[k % len(lists)].append(int(j))
        start = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
        return ids.astype(np.uint32), start, np.concatenate([np.array(l, np.int32) for l in lists]).astype(np.int32)
    extra = [(int(rng.integers(0, nf)), int(rng.integers(0, 1 << 20))) for _ in range(20)] if shared else None
    return dict(kf_desc=kf_desc, f_desc=f_desc, kf_angle=kf_angle, f_angle=f_angle, kf_has_mp=has, kf_nodes=flat(knode), f_nodes=flat(fnode, extra))


def test_search_by_bow():
    """plf_match_bow vs the reference binary's fixture and vs the oracle on random pairs (incl. the serial fallback)"""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    cases = [dict(c, expect=(c["match"], c["nmatches"])) for c in refgen.load_bow_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_bow.json"))]
    for seed, (nkf, nf, nn, sh, ratio, chk) in enumerate([(1000, 1000, 400, False, 0.7, 1), (700, 1200, 90, False, 0.9, 0), (500, 600, 120, True, 0.75, 1), (64, 1, 8, False, 0.7, 1)]):
        c = _bow_random_case(100 + seed, nkf, nf, nn, sh)
        c["nnratio"], c["check"] = ratio, chk
        c["expect"] = orc.search_by_bow(c["kf_desc"], c["f_desc"], c["kf_angle"], c["f_angle"], c["kf_has_mp"], c["kf_nodes"], c["f_nodes"], ratio, chk)
        cases.append(c)
    m = Matcher(max_keypoints=2048, max_mappoints=16, max_batch=8)
    by_setting = {}
    for c in cases:
        by_setting.setdefault((c["nnratio"], c["check"]), []).append(c)
    for (ratio, chk), group in by_setting.items():     # pairs with the same matcher settings go in one batched call
        keep, views = [], []
        for c in group:
            t = [_dev(c[k]) for k in ("kf_desc", "f_desc", "kf_angle", "f_angle", "kf_has_mp")]
            kn = tuple(_dev(x) for x in c["kf_nodes"]); fn = tuple(_dev(x) for x in c["f_nodes"])
            keep.append((t, kn, fn))
            views.append(Matcher.bow_view(t[0], t[1], t[2], t[3], t[4], kn, fn))
        match = torch.full((len(group), 2048), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(len(group), dtype=torch.int32, device="cuda")
        m.SearchByBoW(views, ratio, chk, match, 2048, nm)
        torch.cuda.synchronize()
        for i, c in enumerate(group):
            nf = len(c["f_desc"])
            assert int(nm[i]) == c["expect"][1], (ratio, chk, i)
            assert np.array_equal(match[i, :nf].cpu().numpy(), c["expect"][0]), (ratio, chk, i)
    m.close()


def test_gpu_equals_reference_search_by_projection_lastframe_fixture():
    """HIP matcher vs tests/golden/ref_glue_search_last.json: the result of the reference binary's own
    ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) on the same inputs."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_search_last_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_search_last.json")):
        N = len(c["kps"])
        m = Matcher(max_keypoints=1024)
        dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"]); du = _dev(c["uright"])
        cur = Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), du)
        last = c["last"]
        dl = dict(keys=_kp_tensor(last["keys"]), has_mappoint=_dev(last["has_mappoint"]), outlier=_dev(last["outlier"]), world_pos=_dev(last["world_pos"]),
                  mp_desc=_dev(last["mp_desc"]), obs_positive=_dev(last["obs_positive"]))   # cases 5-6: points without observations (localisation mode)
        match = _dev(c["init"]); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchByProjectionLastFrame(cur, dl, c["pose"], c["th"], c["mono"], c["check"], match, nm)
        torch.cuda.synchronize()
        assert int(nm[0]) == c["nmatches"]
        assert np.array_equal(match.cpu().numpy(), c["match"])
        m.close()


def test_search_by_bow_keyframes():
    """plf_match_bow_kf vs the reference binary's fixture and vs the oracle on random keyframe pairs"""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    cases = []
    for c in refgen.load_bow_kf_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_bow_kf.json")):
        cases.append(dict(c, expect=(c["match"], c["nmatches"])))
    for seed, (n1, n2, nn, ratio, chk) in enumerate([(900, 1000, 300, 0.75, 1), (400, 300, 50, 0.9, 0)]):
        r = _bow_random_case(200 + seed, n1, n2, nn)
        rng = np.random.default_rng(seed)
        c = dict(desc1=r["kf_desc"], desc2=r["f_desc"], angle1=r["kf_angle"], angle2=r["f_angle"], has_mp1=r["kf_has_mp"],
                 has_mp2=(rng.uniform(0, 1, n2) < 0.8).astype(np.uint8), nodes1=r["kf_nodes"], nodes2=r["f_nodes"], nnratio=ratio, check=chk)
        c["expect"] = orc.search_by_bow_kf(c["desc1"], c["desc2"], c["angle1"], c["angle2"], c["has_mp1"], c["has_mp2"], c["nodes1"], c["nodes2"], ratio, chk)
        cases.append(c)
    m = Matcher(max_keypoints=2048, max_mappoints=16, max_batch=4)
    for c in cases:
        t = [_dev(c[k]) for k in ("desc1", "desc2", "angle1", "angle2", "has_mp1", "has_mp2")]
        k1 = tuple(_dev(x) for x in c["nodes1"]); k2 = tuple(_dev(x) for x in c["nodes2"])
        view = Matcher.bow_view(t[0], t[1], t[2], t[3], t[4], k1, k2, f_has_mp=t[5])
        match = torch.full((1, 2048), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchByBoWKeyFrames([view], c["nnratio"], c["check"], match, 2048, nm)
        torch.cuda.synchronize()
        assert int(nm[0]) == c["expect"][1]
        assert np.array_equal(match[0, :len(c["desc1"])].cpu().numpy(), c["expect"][0])
    m.close()


def test_gpu_equals_reference_search_by_projection_reloc_fixture():
    """HIP matcher vs tests/golden/ref_glue_search_reloc.json: the reference binary's own relocalisation
    SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) on the same inputs."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_search_reloc_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_search_reloc.json")):
        N = len(c["kps"])
        m = Matcher(max_keypoints=1024)
        dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"])
        cur = Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), None)
        kf = c["kf"]
        dkf = dict(keys=_kp_tensor(kf["keys"]), valid=_dev(kf["valid"]), world_pos=_dev(kf["world_pos"]), min_dist=_dev(kf["min_dist"]),
                   max_dist=_dev(kf["max_dist"]), mp_desc=_dev(kf["mp_desc"]))
        match = _dev(c["init"]); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchByProjectionKeyFrame(cur, dkf, c["pose"], c["logsf"], c["th"], c["orbdist"], c["check"], match, nm)
        torch.cuda.synchronize()
        assert int(nm[0]) == c["nmatches"]
        assert np.array_equal(match.cpu().numpy(), c["match"])
        m.close()



def test_gpu_equals_reference_fuse_fixture():
    """HIP search half of ORBmatcher::Fuse(KeyFrame*, map points, th) vs tests/golden/ref_glue_fuse.json (the reference binary's own run)."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_fuse_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_fuse.json")):
        N = len(c["kps"]); p = c["pts"]
        m = Matcher(max_keypoints=1024, max_mappoints=1024)
        dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"]); du = _dev(c["uright"])
        kf = Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), du)
        dp = dict(world_pos=_dev(p["xw"]), normal=_dev(p["normal"]), min_dist=_dev(p["min_dist"]), max_dist=_dev(p["max_dist"]), desc=_dev(p["desc"]),
                  valid=_dev(p["valid"]))
        best = torch.full((len(p["valid"]),), -7, dtype=torch.int32, device="cuda"); nf = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.Fuse(kf, c["pose"], dp, c["th"], best, nf)
        torch.cuda.synchronize()
        assert int(nf[0]) == c["nfused"]
        assert np.array_equal(best.cpu().numpy(), c["best_idx"])
        m.close()


def test_gpu_equals_reference_sim3_overloads_fixtures():
    """HIP Fuse(KeyFrame*, Scw, ...) search half and SearchByProjection(KeyFrame*, Scw, ...) vs the reference binary's own runs
    (tests/golden/ref_glue_fuse_sim3.json, ref_glue_search_sim3.json)."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("ref_glue_fuse_sim3.json", "ref_glue_search_sim3.json"):
        for c in refgen.load_sim3_kf_cases(os.path.join(gold, name)):
            N = len(c["kps"]); p = c["pts"]
            m = Matcher(max_keypoints=1024, max_mappoints=1024)
            dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"])
            kf = Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), None)
            dp = dict(world_pos=_dev(p["xw"]), normal=_dev(p["normal"]), min_dist=_dev(p["min_dist"]), max_dist=_dev(p["max_dist"]), desc=_dev(p["desc"]),
                      valid=_dev(p["valid"]))
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            if "fuse" in name:
                out = torch.full((len(p["valid"]),), -7, dtype=torch.int32, device="cuda")
                m.FuseSim3(kf, c["Scw"], c["intr"], dp, c["th"], out, cnt)
            else:
                out = _dev(c["init"])
                m.SearchByProjectionSim3(kf, c["Scw"], c["intr"], dp, int(c["th"]), out, cnt)
            torch.cuda.synchronize()
            assert int(cnt[0]) == c["ret"]
            assert np.array_equal(out.cpu().numpy(), c["out"])
            m.close()


def test_gpu_equals_reference_search_by_sim3_fixture():
    """HIP SearchBySim3 vs tests/golden/ref_glue_sim3.json (the reference binary's own run on two keyframes)."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_sim3_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_sim3.json")):
        N = c["n"]
        m = Matcher(max_keypoints=1024, max_mappoints=1024)
        ds = _dev(c["scale"])
        keep = []
        views = []; pts = []
        for sfx in ("1", "2"):
            dk = _kp_tensor(c["kps" + sfx]); dd = _dev(c["desc" + sfx]); keep += [dk, dd]
            views.append(Matcher.frame_view(N, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), None))
            p = c["pts" + sfx]
            pts.append(dict(world_pos=_dev(p["xw"]), normal=None, min_dist=_dev(p["min_dist"]), max_dist=_dev(p["max_dist"]), desc=_dev(p["desc"]),
                            valid=_dev(p["valid"])))
        match = torch.full((N,), -7, dtype=torch.int32, device="cuda"); nf = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchBySim3(views[0], views[1], c["pose1"], c["pose2"], c["s12"], c["R12"], c["t12"], c["th"], pts[0], pts[1], match, nf)
        torch.cuda.synchronize()
        assert int(nf[0]) == c["nfound"]
        assert np.array_equal(match.cpu().numpy(), c["match12"])
        m.close()


def test_gpu_equals_reference_search_for_triangulation_fixture():
    """HIP SearchForTriangulation vs tests/golden/ref_glue_triangulation.json (the reference binary's own run)."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_triangulation_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_triangulation.json")):
        N = c["n"]
        m = Matcher(max_keypoints=1024, max_mappoints=1024)
        kfs = []
        for sfx in ("1", "2"):
            kfs.append(dict(keys=_kp_tensor(c["kps" + sfx]), uright=_dev(c["uright" + sfx]), desc=_dev(c["desc" + sfx]), has_mp=_dev(c["has_mp" + sfx]),
                            nodes=tuple(_dev(a) for a in c["nodes" + sfx])))
        kfs[1]["scale_factors"] = _dev(c["scale"]); kfs[1]["level_sigma2"] = _dev(c["sigma2"])
        pose2 = dict(Rcw=c["R2w"], tcw=c["t2w"], Ow=np.zeros(3, np.float32), fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"], bf=0.0, log_scale_factor=1.0,
                     inv_sigma2=np.ones(8, np.float32))
        match = torch.full((N,), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.SearchForTriangulation(kfs[0], kfs[1], c["F12"], c["Ow1"], pose2, c["only_stereo"], c["check"], match, nm)
        torch.cuda.synchronize()
        assert int(nm[0]) == c["nmatches"]
        assert np.array_equal(match.cpu().numpy(), c["match12"])
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,nk,m", [(1, 300, 64), (2, 2000, 5000), (3, 4000, 12000), (4, 50, 3000)])
def test_keyframe_projection_family_random_scenes(seed, nk, m):
    """Fuse (both overloads) and the Scw SearchByProjection on random scenes, HIP vs the oracle (itself pinned to the reference binary)."""
    import kfgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    c = kfgen.keyframe_scene(seed, nk, m, sim3_scale=1.0 + 0.03 * seed)
    p = c["pts"]
    mt = Matcher(max_keypoints=4096, max_mappoints=16384)
    dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"]); du = _dev(c["uright"])
    kf = Matcher.frame_view(nk, dk, dd, ds, c["bounds"], du)
    dp = dict(world_pos=_dev(p["xw"]), normal=_dev(p["normal"]), min_dist=_dev(p["min_dist"]), max_dist=_dev(p["max_dist"]), desc=_dev(p["desc"]), valid=_dev(p["valid"]))
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for th in (3.0, 7.5):
        best = torch.full((m,), -7, dtype=torch.int32, device="cuda")
        mt.Fuse(kf, c["pose"], dp, th, best, cnt); torch.cuda.synchronize()
        eb, ed, en = orc.fuse(c["kps"], c["desc"], c["uright"], c["scale"], c["bounds"], c["pose"], p, th)
        assert int(cnt[0]) == en and en > 0 and np.array_equal(best.cpu().numpy(), eb)
        mt.FuseSim3(kf, c["Scw"], c["intr"], dp, th, best, cnt); torch.cuda.synchronize()
        eb, en = orc.fuse_sim3(c["kps"], c["desc"], c["scale"], c["bounds"], c["Scw"], c["intr"], p, th)
        assert int(cnt[0]) == en and np.array_equal(best.cpu().numpy(), eb)
        match = _dev(c["init"])
        mt.SearchByProjectionSim3(kf, c["Scw"], c["intr"], dp, int(th), match, cnt); torch.cuda.synchronize()
        em, en = orc.search_by_projection_sim3(c["kps"], c["desc"], c["scale"], c["bounds"], c["Scw"], c["intr"], p, int(th), c["init"])
        assert int(cnt[0]) == en and np.array_equal(match.cpu().numpy(), em)
    mt.close()


@pytest.mark.gpu
def test_keyframe_projection_family_edge_cases():
    """no map points, no key points, nothing valid, argument errors"""
    import kfgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    from rgbd_pl_slam_amd._lib import PlfError
    c = kfgen.keyframe_scene(9, 200, 100, sim3_scale=1.0)
    p = c["pts"]
    mt = Matcher(max_keypoints=256, max_mappoints=128)
    dk = _kp_tensor(c["kps"]); dd = _dev(c["desc"]); ds = _dev(c["scale"]); du = _dev(c["uright"])
    dp = dict(world_pos=_dev(p["xw"]), normal=_dev(p["normal"]), min_dist=_dev(p["min_dist"]), max_dist=_dev(p["max_dist"]), desc=_dev(p["desc"]), valid=_dev(p["valid"]))
    cnt = torch.full((1,), 77, dtype=torch.int32, device="cuda"); best = torch.full((100,), -7, dtype=torch.int32, device="cuda")
    empty_kf = Matcher.frame_view(0, dk, dd, ds, c["bounds"], du)
    mt.Fuse(empty_kf, c["pose"], dp, 3.0, best, cnt); torch.cuda.synchronize()
    assert int(cnt[0]) == 0 and (best.cpu().numpy() == -1).all()
    kf = Matcher.frame_view(200, dk, dd, ds, c["bounds"], du)
    none = dict(dp); none["valid"] = torch.zeros(100, dtype=torch.uint8, device="cuda")
    mt.FuseSim3(kf, c["Scw"], c["intr"], none, 3.0, best, cnt); torch.cuda.synchronize()
    assert int(cnt[0]) == 0 and (best.cpu().numpy() == -1).all()
    zero = {k: v[:0] for k, v in dp.items()}
    match = _dev(c["init"]); before = match.clone()
    mt.SearchByProjectionSim3(kf, c["Scw"], c["intr"], zero, 3, match, cnt); torch.cuda.synchronize()
    assert int(cnt[0]) == 0 and torch.equal(match, before)
    big = kfgen.keyframe_scene(10, 200, 300)
    bp = big["pts"]
    dbig = dict(world_pos=_dev(bp["xw"]), normal=_dev(bp["normal"]), min_dist=_dev(bp["min_dist"]), max_dist=_dev(bp["max_dist"]), desc=_dev(bp["desc"]), valid=_dev(bp["valid"]))
    with pytest.raises(PlfError):   # more map points than the handle was sized for
        mt.Fuse(kf, c["pose"], dbig, 3.0, torch.zeros(300, dtype=torch.int32, device="cuda"), cnt)
    bad = dict(c["pose"]); bad["log_scale_factor"] = 0.0
    with pytest.raises(PlfError):
        mt.Fuse(kf, bad, dp, 3.0, best, cnt)
    mt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,tz,s12", [(1, 400, 0.05, 1.0), (2, 2500, 0.4, 1.05), (3, 4000, 0.2, 0.96)])
def test_two_keyframe_overloads_random_scenes(seed, n, tz, s12):
    """SearchBySim3 and SearchForTriangulation on random two-keyframe scenes, HIP vs the oracle (itself pinned to the reference binary)."""
    import kfgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import Matcher
    c = kfgen.two_keyframe_scene(seed, n, nodes=max(20, n // 8), s12=s12, tz=tz)
    mt = Matcher(max_keypoints=4096, max_mappoints=4096)
    ds = _dev(c["scale"])
    keep, views, pts, kfs = [], [], [], []
    for sfx in ("1", "2"):
        dk = _kp_tensor(c["kps" + sfx]); dd = _dev(c["desc" + sfx]); keep += [dk, dd]
        views.append(Matcher.frame_view(n, dk, dd, ds, (0.0, 0.0, 640.0, 480.0), None))
        p = c["pts" + sfx]
        pts.append(dict(world_pos=_dev(p["xw"]), normal=None, min_dist=_dev(p["min_dist"]), max_dist=_dev(p["max_dist"]), desc=_dev(p["desc"]), valid=_dev(p["valid"])))
        kfs.append(dict(keys=dk, uright=_dev(c["uright" + sfx]), desc=dd, has_mp=_dev(c["has_mp" + sfx]), nodes=tuple(_dev(a) for a in c["nodes" + sfx])))
    kfs[1]["scale_factors"] = ds; kfs[1]["level_sigma2"] = _dev(c["sigma2"])
    match = torch.full((n,), -7, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    c["th"] = 7.5
    mt.SearchBySim3(views[0], views[1], c["pose1"], c["pose2"], c["s12"], c["R12"], c["t12"], c["th"], pts[0], pts[1], match, cnt); torch.cuda.synchronize()
    em, en = orc.search_by_sim3(c)
    assert en > n // 20 and int(cnt[0]) == en and np.array_equal(match.cpu().numpy(), em)
    for only_stereo, check in ((0, 1), (1, 0), (0, 0)):
        c["only_stereo"] = only_stereo; c["check"] = check
        mt.SearchForTriangulation(kfs[0], kfs[1], c["F12"], c["Ow1"], c["pose2"], only_stereo, check, match, cnt); torch.cuda.synchronize()
        em, en = orc.search_for_triangulation(c)
        assert en > 0 and int(cnt[0]) == en and np.array_equal(match.cpu().numpy(), em)
    mt.close()


@pytest.mark.gpu
def test_gpu_equals_reference_assign_features_to_grid_fixture():
    """k_build_grid vs tests/golden/ref_glue_grid.json (Frame::AssignFeaturesToGrid run from the reference binary)."""
    import os
    import refgen
    import kfgen
    _need_gpu()
    from rgbd_pl_slam_amd import Matcher
    for c in refgen.load_grid_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_grid.json")):
        n = len(c["x"])
        kps = np.zeros(n, kfgen.KP_DTYPE); kps["x"] = c["x"]; kps["y"] = c["y"]
        m = Matcher(max_keypoints=2048)
        dk = _kp_tensor(kps); dd = _dev(np.zeros((n, 32), np.uint8)); ds = _dev(np.ones(8, np.float32))
        cs, ci = m.AssignFeaturesToGrid(Matcher.frame_view(n, dk, dd, ds, c["bounds"], None))
        assert np.array_equal(cs, c["cell_start"]) and np.array_equal(ci, c["cell_idx"])
        m.close()
