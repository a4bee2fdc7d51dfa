"""Oracle restatements of the Frame tail / ingest / frustum rows: closed-form sanity checks (no GPU)."""
import numpy as np

import orc


def test_gray_weights_and_depth_scale():
    rgb = np.zeros((2, 3, 3), np.uint8)
    rgb[0, 0] = (255, 255, 255); rgb[0, 1] = (255, 0, 0); rgb[0, 2] = (0, 255, 0); rgb[1, 0] = (0, 0, 255)
    g = orc.rgb_to_gray(rgb, bgr=False)
    assert g[0, 0] == 255 and g[0, 1] == 76 and g[0, 2] == 150 and g[1, 0] == 29      # 0.299 / 0.587 / 0.114
    assert np.array_equal(orc.rgb_to_gray(rgb[:, :, ::-1], bgr=True), g)
    d = np.array([[0, 5000, 12345]], np.uint16)
    f = orc.depth_to_float(d, np.float32(1.0) / np.float32(5000.0))
    assert f[0, 0] == 0 and abs(f[0, 1] - 1.0) < 1e-6 and abs(f[0, 2] - 2.469) < 1e-5


def test_undistort_roundtrip_and_identity():
    kps = np.zeros(5, orc.KP_DTYPE)
    kps["x"] = [100, 320, 600, 10, 318.64304]; kps["y"] = [50, 240, 400, 470, 255.313989]
    cam = [517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    depth = np.full((480, 640), 2.0, np.float32)
    un, ur, kd = orc.frame_tail(kps, depth, cam, 40.0)
    # the principal point is a fixed point of the distortion model
    assert abs(un["x"][4] - kps["x"][4]) < 1e-3 and abs(un["y"][4] - kps["y"][4]) < 1e-3
    # re-distorting the undistorted point gives back the original pixel (5 iterations converge for TUM1)
    x = (un["x"].astype(np.float64) - cam[2]) / cam[0]; y = (un["y"].astype(np.float64) - cam[3]) / cam[1]
    r2 = x * x + y * y
    cd = 1 + cam[4] * r2 + cam[5] * r2 ** 2 + cam[8] * r2 ** 3
    xd = x * cd + 2 * cam[6] * x * y + cam[7] * (r2 + 2 * x * x); yd = y * cd + cam[6] * (r2 + 2 * y * y) + 2 * cam[7] * x * y
    assert np.allclose(xd * cam[0] + cam[2], kps["x"], atol=0.05) and np.allclose(yd * cam[1] + cam[3], kps["y"], atol=0.05)
    assert np.allclose(ur, un["x"] - 40.0 / 2.0) and np.all(kd == 2.0)
    un0, _, _ = orc.frame_tail(kps, depth, cam[:4] + [0, 0, 0, 0, 0], 40.0)
    assert np.array_equal(un0["x"], kps["x"]) and np.array_equal(un0["y"], kps["y"])


def test_frustum_simple_geometry():
    xw = np.array([[0, 0, 2], [0, 0, -1], [50, 0, 2], [0, 0, 30]], np.float32)
    nrm = np.array([[0, 0, 1]] * 4, np.float32)
    r = orc.is_in_frustum(xw, nrm, [0.5] * 4, [4.0] * 4, np.eye(3), [0, 0, 0], [0, 0, 0], [500, 500, 320, 240], (0, 0, 640, 480), 40.0,
                          float(np.log(np.float32(1.2))), 8, 0.5)
    assert r["in_view"].tolist() == [1, 0, 0, 0]            # behind camera / outside image / too far
    assert r["proj_x"][0] == 320 and r["proj_y"][0] == 240 and abs(r["proj_xr"][0] - (320 - 20)) < 1e-4
    assert r["view_cos"][0] == 1.0 and r["level"][0] == int(np.ceil(np.log(4.0 / 2.0) / np.log(1.2)))


def test_line_tail_is_the_point_tail_of_both_end_points():
    """orc_line_tail (Frame.h:207-211, :267; no body in the snapshot) == the pinned point routines applied to the end points"""
    rng = np.random.default_rng(5)
    n = 64
    kls = np.zeros(n, orc.KL_DTYPE)
    for f, hi in (("startPointX", 639), ("startPointY", 479), ("endPointX", 639), ("endPointY", 479)):
        kls[f] = rng.uniform(0, hi, n).astype(np.float32)
    kls["endPointX"][0] = 700.0; kls["startPointY"][1] = -3.0         # outside the image: no depth
    kls["lineLength"] = rng.uniform(5, 100, n).astype(np.float32); kls["class_id"] = np.arange(n); kls["octave"] = 0
    cam = [517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    depth = rng.uniform(0.3, 6.0, (480, 640)).astype(np.float32)
    depth[rng.random((480, 640)) < 0.2] = 0.0
    un, urs, ure, ds, de = orc.line_tail(kls, depth, cam, 40.0)
    for a, b, ur, dd in (("startPointX", "startPointY", urs, ds), ("endPointX", "endPointY", ure, de)):
        kps = np.zeros(n, orc.KP_DTYPE); kps["x"] = kls[a]; kps["y"] = kls[b]
        pun, pur, pkd = orc.frame_tail(kps, depth, cam, 40.0)
        assert np.array_equal(un[a].view(np.uint32), pun["x"].view(np.uint32)) and np.array_equal(un[b].view(np.uint32), pun["y"].view(np.uint32))
        assert np.array_equal(ur.view(np.uint32), pur.view(np.uint32)) and np.array_equal(dd.view(np.uint32), pkd.view(np.uint32))
    assert ure[0] == -1 and de[0] == -1 and urs[1] == -1 and ds[1] == -1 and (ds == 0).sum() == 0 and (ds == -1).sum() > 3
    for f in ("angle", "class_id", "octave", "pt_x", "pt_y", "response", "size", "lineLength", "numOfPixels", "sPointInOctaveX", "ePointInOctaveY"):
        assert np.array_equal(un[f], kls[f])
    un0, urs0, _, ds0, _ = orc.line_tail(kls, None, cam[:4] + [0, 0, 0, 0, 0], 40.0)       # no distortion, monocular
    assert un0.tobytes() == kls.tobytes() and np.all(urs0 == -1) and np.all(ds0 == -1)


def _line_scene(rng, m):
    s = rng.uniform(-3, 3, (m, 3)).astype(np.float32); s[:, 2] = rng.uniform(-1, 6, m)
    e = (s + rng.normal(0, 0.4, (m, 3))).astype(np.float32)
    nrm = rng.normal(0, 1, (m, 3)).astype(np.float32); nrm[:, 2] += 1.5
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    dmax = rng.uniform(2, 12, m).astype(np.float32); dmin = (dmax / rng.uniform(2, 6, m)).astype(np.float32)
    a = 0.1
    Rcw = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    tcw = np.array([0.05, -0.02, 0.1], np.float32)
    Ow = (-Rcw.T @ tcw).astype(np.float32)
    return np.concatenate([s, e], axis=1), nrm, dmin, dmax, Rcw, tcw, Ow


def test_line_frustum_reduces_to_the_point_routine():
    """a degenerate segment (start == end) must give exactly the pinned MapPoint result; a real segment needs both end points in view"""
    rng = np.random.default_rng(9)
    m = 4000
    xw6, nrm, dmin, dmax, Rcw, tcw, Ow = _line_scene(rng, m)
    cam4 = [517.306408, 516.469215, 318.643040, 255.313989]; bounds = (-20.0, -15.0, 660.0, 495.0)
    logsf = float(np.log(np.float32(1.2)).astype(np.float32))
    deg = xw6.copy(); deg[:, 3:] = deg[:, :3]
    rl = orc.is_in_frustum_line(deg, nrm, dmin, dmax, Rcw, tcw, Ow, cam4, bounds, 40.0, logsf, 8, 0.5)
    rp = orc.is_in_frustum(deg[:, :3], nrm, dmin, dmax, Rcw, tcw, Ow, cam4, bounds, 40.0, logsf, 8, 0.5)
    assert np.array_equal(rl["in_view"], rp["in_view"]) and 100 < rp["in_view"].sum() < m
    sel = rp["in_view"] == 1
    for a, b in (("x1", "proj_x"), ("y1", "proj_y"), ("x1r", "proj_xr"), ("x2", "proj_x"), ("y2", "proj_y"), ("x2r", "proj_xr"), ("view_cos", "view_cos")):
        assert np.array_equal(rl[a][sel].view(np.uint32), rp[b][sel].view(np.uint32)), a
    assert np.array_equal(rl["level"][sel], rp["level"][sel])
    r = orc.is_in_frustum_line(xw6, nrm, dmin, dmax, Rcw, tcw, Ow, cam4, bounds, 40.0, logsf, 8, 0.5)
    ps = orc.is_in_frustum(xw6[:, :3], nrm, np.zeros(m), np.full(m, 1e9), Rcw, tcw, Ow, cam4, bounds, 40.0, logsf, 8, -2.0)
    pe = orc.is_in_frustum(xw6[:, 3:], nrm, np.zeros(m), np.full(m, 1e9), Rcw, tcw, Ow, cam4, bounds, 40.0, logsf, 8, -2.0)
    assert not np.any(r["in_view"] & ~(ps["in_view"] & pe["in_view"])) and 50 < r["in_view"].sum() < (ps["in_view"] & pe["in_view"]).sum()
    sel = r["in_view"] == 1
    assert np.array_equal(r["x1"][sel].view(np.uint32), ps["proj_x"][sel].view(np.uint32)) and np.array_equal(r["y2"][sel].view(np.uint32), pe["proj_y"][sel].view(np.uint32))
