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
