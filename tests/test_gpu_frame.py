"""GPU parity of the "next" rows: RGB-D ingest, Frame tail, frustum projection."""
import numpy as np
import pytest

import orc
from conftest import gpu_available

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_rgbd_ingest():
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
    d16 = rng.integers(0, 65536, (3, 480, 640), dtype=np.uint16); d16[rng.uniform(0, 1, d16.shape) < 0.05] = 0
    for bgr in (True, False):
        g = torch.zeros((3, 480, 640), dtype=torch.uint8, device="cuda")
        drgb = _dev(rgb)
        frame.rgb_to_gray(drgb, g, bgr_order=bgr)
        torch.cuda.synchronize()
        for f in range(3):
            assert np.array_equal(g[f].cpu().numpy(), orc.rgb_to_gray(rgb[f], bgr))
    out = torch.zeros((3, 480, 640), dtype=torch.float32, device="cuda")
    dd = torch.from_numpy(d16.view(np.int16)).cuda()      # same bits; torch lacks uint16 kernels in some builds
    factor = np.float32(1.0) / np.float32(5000.0)          # 1.0f / DepthMapFactor (TUM1.yaml:35)
    frame.depth_to_float(dd, out, float(factor))
    torch.cuda.synchronize()
    for f in range(3):
        assert np.array_equal(out[f].cpu().numpy().view(np.uint32), orc.depth_to_float(d16[f], factor).view(np.uint32))


def test_frame_tail_undistort_and_stereo():
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame
    from rgbd_pl_slam_amd.synth import synth_frame
    from rgbd_pl_slam_amd._lib import KP_DTYPE
    gray, d16 = synth_frame(3, with_depth=True)
    r = orc.orb_extract(gray)
    kps = r["kps"]; n = len(kps)
    depth = orc.depth_to_float(d16, np.float32(1.0) / np.float32(5000.0))
    c = frame.TUM1
    cam9 = [c[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]
    for dist in (True, False):
        cc = dict(c)
        if not dist:
            cc.update(k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0)
        cam = frame.camera(**cc)
        c9 = list(cam9) if dist else cam9[:4] + [0, 0, 0, 0, 0]
        un, ur, kd = orc.frame_tail(kps, depth, c9, c["bf"])
        dk = torch.from_numpy(np.frombuffer(kps.tobytes(), np.uint8).copy()).cuda()
        dun = torch.zeros(n * 28, dtype=torch.uint8, device="cuda"); dur = torch.zeros(n, dtype=torch.float32, device="cuda")
        dkd = torch.zeros(n, dtype=torch.float32, device="cuda")
        ddepth = _dev(depth)
        frame.frame_tail(dk, n, 1, n, ddepth, 640, 480, cam, dun, dur, dkd)
        torch.cuda.synchronize()
        gun = np.frombuffer(dun.cpu().numpy().tobytes(), KP_DTYPE)
        for f in ("x", "y", "angle", "size", "response"):
            assert np.array_equal(gun[f].view(np.uint32), un[f].view(np.uint32)), f
        assert np.array_equal(dur.cpu().numpy().view(np.uint32), ur.view(np.uint32))
        assert np.array_equal(dkd.cpu().numpy().view(np.uint32), kd.view(np.uint32))
        if dist:
            assert np.abs(gun["x"] - kps["x"]).max() > 0.5     # the TUM1 distortion really moves points
        assert (ur > 0).sum() > 0.8 * n


def test_frustum_projection():
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame
    rng = np.random.default_rng(4)
    m = 20000
    xw = rng.uniform(-3, 3, (m, 3)).astype(np.float32); xw[:, 2] = rng.uniform(-1, 6, m)
    nrm = rng.normal(0, 1, (m, 3)).astype(np.float32); nrm[:, 2] += 1.5
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    dmax = rng.uniform(2, 12, m).astype(np.float32); dmin = (dmax / rng.uniform(2, 6, m)).astype(np.float32)
    a = 0.1
    Rcw = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    tcw = np.array([0.05, -0.02, 0.1], np.float32)
    Ow = (-Rcw.T @ tcw).astype(np.float32)
    c = frame.TUM1
    bounds = (-20.0, -15.0, 660.0, 495.0)
    logsf = float(np.log(np.float32(1.2)).astype(np.float32))
    ref = orc.is_in_frustum(xw, nrm, dmin, dmax, Rcw, tcw, Ow, [c["fx"], c["fy"], c["cx"], c["cy"]], bounds, c["bf"], logsf, 8, 0.5)
    assert 500 < ref["in_view"].sum() < m
    out = dict(proj_x=torch.zeros(m, device="cuda"), proj_y=torch.zeros(m, device="cuda"), proj_xr=torch.zeros(m, device="cuda"),
               level=torch.zeros(m, dtype=torch.int32, device="cuda"), view_cos=torch.zeros(m, device="cuda"),
               in_view=torch.zeros(m, dtype=torch.uint8, device="cuda"))
    keep = [_dev(xw), _dev(nrm), _dev(dmin), _dev(dmax)]
    frame.frustum_points(keep[0], keep[1], keep[2], keep[3], dict(Rcw=Rcw, tcw=tcw, Ow=Ow), frame.camera(**c), bounds, logsf, 8, 0.5, out)
    torch.cuda.synchronize()
    iv = out["in_view"].cpu().numpy()
    assert np.array_equal(iv, ref["in_view"])
    sel = iv == 1
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
        assert np.array_equal(out[k].cpu().numpy()[sel].view(np.uint32), ref[k][sel].view(np.uint32)), k
    assert np.array_equal(out["level"].cpu().numpy()[sel], ref["level"][sel])


def test_gpu_equals_reference_is_in_frustum_fixture():
    """plf_frustum_points vs tests/golden/ref_glue_frustum.json (outputs of the reference binary's Frame::isInFrustum)"""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame
    for c in refgen.load_frustum_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_frustum.json")):
        m = len(c["xw"]); cam = c["cam"]
        out = dict(proj_x=torch.zeros(m, device="cuda"), proj_y=torch.zeros(m, device="cuda"), proj_xr=torch.zeros(m, device="cuda"),
                   level=torch.zeros(m, dtype=torch.int32, device="cuda"), view_cos=torch.zeros(m, device="cuda"),
                   in_view=torch.zeros(m, dtype=torch.uint8, device="cuda"))
        keep = [_dev(c["xw"]), _dev(c["normal"]), _dev(c["dmin"]), _dev(c["dmax"])]
        camd = dict(frame.TUM1); camd.update(fx=float(cam[0]), fy=float(cam[1]), cx=float(cam[2]), cy=float(cam[3]), bf=float(cam[4]))
        frame.frustum_points(keep[0], keep[1], keep[2], keep[3], dict(Rcw=c["Rcw"], tcw=c["tcw"], Ow=c["Ow"]), frame.camera(**camd), (0.0, 0.0, 640.0, 480.0),
                             float(cam[5]), 8, float(cam[6]), out)
        torch.cuda.synchronize()
        iv = out["in_view"].cpu().numpy()
        assert np.array_equal(iv, c["in_view"])
        sel = iv == 1
        for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
            assert np.array_equal(out[k].cpu().numpy()[sel].view(np.uint32), c[k][sel].view(np.uint32)), k
        assert np.array_equal(out["level"].cpu().numpy()[sel], c["level"][sel])



def test_gpu_equals_reference_undistort_fixture():
    """plf_frame_tail vs tests/golden/ref_glue_undistort.json (Frame::UndistortKeyPoints run from the reference binary)."""
    import os
    import refgen
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame
    from rgbd_pl_slam_amd._lib import KP_DTYPE
    for c in refgen.load_undistort_cases(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_undistort.json")):
        n = c["n"]
        kps = np.zeros(n, KP_DTYPE); kps["x"] = c["x"]; kps["y"] = c["y"]; kps["octave"] = 2
        cam = frame.camera(**dict(zip(("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"), [float(v) for v in c["cam"]])), bf=40.0)
        dk = torch.from_numpy(np.frombuffer(kps.tobytes(), np.uint8).copy()).cuda()
        dun = torch.zeros(n * 28, dtype=torch.uint8, device="cuda"); dur = torch.zeros(n, dtype=torch.float32, device="cuda"); dkd = torch.zeros(n, dtype=torch.float32, device="cuda")
        frame.frame_tail(dk, n, 1, n, torch.zeros((480, 640), dtype=torch.float32, device="cuda"), 640, 480, cam, dun, dur, dkd)
        torch.cuda.synchronize()
        gun = np.frombuffer(dun.cpu().numpy().tobytes(), KP_DTYPE)
        assert np.array_equal(gun["x"].view(np.uint32), c["x_un"].view(np.uint32)) and np.array_equal(gun["y"].view(np.uint32), c["y_un"].view(np.uint32))
        assert (gun["octave"] == 2).all()


def test_frame_line_tail_batch():
    """plf_frame_line_tail on the device outputs of the batched line extractor == oracle (bit-exact), with and without distortion / depth"""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame, LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    from rgbd_pl_slam_amd._lib import KL_DTYPE
    B = 3
    fr = [synth_frame(40 + i, with_depth=True) for i in range(B)]
    gray = np.stack([f[0] for f in fr]); d16 = np.stack([f[1] for f in fr])
    depth = np.stack([orc.depth_to_float(d, np.float32(1.0) / np.float32(5000.0)) for d in d16])
    ls = LineSegment(nlines=100, max_batch=B)
    res = ls.extract_batch(gray)
    cap = 100
    host = np.zeros((B, cap), KL_DTYPE); counts = np.zeros(B, np.int32)
    for f in range(B):
        kl = res[f][0]; counts[f] = len(kl); host[f, :len(kl)] = kl
    dl = torch.from_numpy(np.frombuffer(host.tobytes(), np.uint8).copy()).cuda()
    dn = torch.from_numpy(counts).cuda()
    ddepth = _dev(depth)
    c = frame.TUM1
    cam9 = [c[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]
    for dist, with_depth in ((True, True), (False, True), (True, False)):
        cc = dict(c)
        if not dist:
            cc.update(k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0)
        cam = frame.camera(**cc)
        c9 = list(cam9) if dist else cam9[:4] + [0, 0, 0, 0, 0]
        dun = torch.zeros(B * cap * 68, dtype=torch.uint8, device="cuda")
        outs = [torch.full((B, cap), 7.0, dtype=torch.float32, device="cuda") for _ in range(4)]
        frame.frame_line_tail(dl, dn, B, cap, ddepth if with_depth else None, 640, 480, cam, dun, *outs)
        torch.cuda.synchronize()
        gun = np.frombuffer(dun.cpu().numpy().tobytes(), KL_DTYPE).reshape(B, cap)
        moved = 0.0
        for f in range(B):
            n = counts[f]
            un, urs, ure, ds, de = orc.line_tail(host[f, :n], depth[f] if with_depth else None, c9, c["bf"])
            assert gun[f, :n].tobytes() == un.tobytes()
            for g, o in zip(outs, (urs, ure, ds, de)):
                gg = g[f].cpu().numpy()
                assert np.array_equal(gg[:n].view(np.uint32), o.view(np.uint32))
                assert np.all(gg[n:] == 7.0)                                        # nothing written past the frame's count
            moved = max(moved, float(np.abs(un["startPointX"] - host[f, :n]["startPointX"]).max()))
            if with_depth:
                assert (ds > 0).sum() > n // 2
        assert (moved > 0.5) == dist


def test_frustum_lines():
    """plf_frustum_lines == oracle bit for bit; its outputs are the plf_mapline_view fields LSDmatcher::SearchByProjection reads"""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import frame
    from test_oracle_frame import _line_scene
    rng = np.random.default_rng(11)
    m = 30001
    xw6, nrm, dmin, dmax, Rcw, tcw, Ow = _line_scene(rng, m)
    c = frame.TUM1
    bounds = (-20.0, -15.0, 660.0, 495.0)
    logsf = float(np.log(np.float32(1.2)).astype(np.float32))
    ref = orc.is_in_frustum_line(xw6, nrm, dmin, dmax, Rcw, tcw, Ow, [c["fx"], c["fy"], c["cx"], c["cy"]], bounds, c["bf"], logsf, 8, 0.5)
    assert 500 < ref["in_view"].sum() < m
    for with_right in (True, False):
        out = {k: torch.zeros(m, device="cuda") for k in ("x1", "y1", "x2", "y2", "view_cos")}
        out["x1r"] = torch.zeros(m, device="cuda") if with_right else None
        out["x2r"] = torch.zeros(m, device="cuda") if with_right else None
        out["level"] = torch.zeros(m, dtype=torch.int32, device="cuda"); out["in_view"] = torch.full((m,), 9, dtype=torch.uint8, device="cuda")
        keep = [_dev(xw6), _dev(nrm), _dev(dmin), _dev(dmax)]
        frame.frustum_lines(keep[0], keep[1], keep[2], keep[3], dict(Rcw=Rcw, tcw=tcw, Ow=Ow), frame.camera(**c), bounds, logsf, 8, 0.5, out)
        torch.cuda.synchronize()
        iv = out["in_view"].cpu().numpy()
        assert np.array_equal(iv, ref["in_view"])
        sel = iv == 1
        for k in ("x1", "y1", "x2", "y2", "view_cos") + (("x1r", "x2r") if with_right else ()):
            assert np.array_equal(out[k].cpu().numpy()[sel].view(np.uint32), ref[k][sel].view(np.uint32)), k
        assert np.array_equal(out["level"].cpu().numpy()[sel], ref["level"][sel])
