"""Host-side mirror of the Frame stages around the extractor / matcher: RGB-D ingest
(Tracking::GrabImageRGBD), Frame::UndistortKeyPoints + ComputeStereoFromRGBD, Frame::isInFrustum.  Device tensors in/out."""
import ctypes as C

import numpy as np

from . import _lib as L


class Camera(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")]


class FrustumPose(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3)]


TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, k1=0.262383, k2=-0.953104, p1=-0.005358, p2=0.002628,
            k3=1.163314, bf=40.0)   # Examples/RGB-D/TUM1.yaml:8-32


def camera(**kw):
    c = Camera()
    for k, v in kw.items():
        setattr(c, k, float(v))
    return c


def _stream(s):
    return C.c_void_p(s) if s else None


def rgb_to_gray(d_rgb, d_gray, bgr_order=True, device=0, stream=None):
    """d_rgb: (B,H,W,3) uint8 device tensor -> d_gray (B,H,W) uint8"""
    B, H, W, _ = d_rgb.shape
    L.check(L.lib().plf_rgb_to_gray(L.vp(d_rgb), B, W, H, C.c_ssize_t(3 * W), C.c_ssize_t(3 * W * H), int(bgr_order), L.vp(d_gray),
                                    C.c_ssize_t(W), C.c_ssize_t(W * H), device, _stream(stream)), "plf_rgb_to_gray")


def depth_to_float(d_depth16, d_out, factor, device=0, stream=None):
    B, H, W = d_depth16.shape
    L.check(L.lib().plf_depth_to_float(L.vp(d_depth16), B, W, H, C.c_ssize_t(W), C.c_ssize_t(W * H), C.c_float(factor), L.vp(d_out), device,
                                       _stream(stream)), "plf_depth_to_float")


def frame_tail(d_keys, n, n_frames, kp_stride, d_depth, W, H, cam, d_keys_un, d_uright, d_kdepth, device=0, stream=None):
    """n: int (host count, single frame) or a device int32 tensor with one count per frame"""
    n_dev, n_host = (None, int(n)) if isinstance(n, int) else (L.vp(n), 0)
    L.check(L.lib().plf_frame_tail(L.vp(d_keys), n_dev, n_host, n_frames, kp_stride, L.vp(d_depth) if d_depth is not None else None, W, H,
                                   C.byref(cam), L.vp(d_keys_un), L.vp(d_uright), L.vp(d_kdepth), device, _stream(stream)), "plf_frame_tail")


def frame_line_tail(d_lines, n, n_frames, line_stride, d_depth, W, H, cam, d_lines_un, d_ur_start, d_ur_end, d_depth_start, d_depth_end, device=0,
                    stream=None):
    """Frame::UndistortKeyLines + mvuRightLineStart/End, mvDepthLineStart/End (include/Frame.h:207-211, :267) for the KeyLines of n_frames frames
    (layout of LineSegment.extract_batch's device outputs).  n: int or a device int32 tensor with one count per frame."""
    n_dev, n_host = (None, int(n)) if isinstance(n, int) else (L.vp(n), 0)
    opt = lambda t: L.vp(t) if t is not None else None
    L.check(L.lib().plf_frame_line_tail(L.vp(d_lines), n_dev, n_host, n_frames, line_stride, opt(d_depth), W, H, C.byref(cam), L.vp(d_lines_un),
                                        opt(d_ur_start), opt(d_ur_end), opt(d_depth_start), opt(d_depth_end), device, _stream(stream)),
            "plf_frame_line_tail")


def frustum_points(d_xw, d_normal, d_min, d_max, pose, cam, bounds, log_scale_factor, nlevels, cos_limit, out, device=0, stream=None):
    """out: dict of device tensors proj_x, proj_y, proj_xr, level, view_cos, in_view (the plf_mappoint_view fields)"""
    pp = FrustumPose()
    for name in ("Rcw", "tcw", "Ow"):
        getattr(pp, name)[:] = np.asarray(pose[name], np.float32).ravel().tolist()
    m = int(d_xw.shape[0])
    L.check(L.lib().plf_frustum_points(L.vp(d_xw), L.vp(d_normal), L.vp(d_min), L.vp(d_max), m, C.byref(pp), C.byref(cam), C.c_float(bounds[0]),
                                       C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), C.c_float(log_scale_factor), nlevels,
                                       C.c_float(cos_limit), L.vp(out["proj_x"]), L.vp(out["proj_y"]), L.vp(out["proj_xr"]), L.vp(out["level"]),
                                       L.vp(out["view_cos"]), L.vp(out["in_view"]), device, _stream(stream)), "plf_frustum_points")


def frustum_lines(d_xw6, d_normal, d_min, d_max, pose, cam, bounds, log_scale_factor, nlevels, cos_limit, out, device=0, stream=None):
    """Frame::isInFrustum(MapLine*) (include/Frame.h:107) for m map lines (d_xw6: m x 6 floats, start xyz then end xyz).
    out: dict of device tensors x1, y1, x1r, x2, y2, x2r, level, view_cos, in_view (the plf_mapline_view fields; x1r / x2r optional)"""
    pp = FrustumPose()
    for name in ("Rcw", "tcw", "Ow"):
        getattr(pp, name)[:] = np.asarray(pose[name], np.float32).ravel().tolist()
    m = int(d_xw6.shape[0])
    opt = lambda k: L.vp(out[k]) if out.get(k) is not None else None
    L.check(L.lib().plf_frustum_lines(L.vp(d_xw6), L.vp(d_normal), L.vp(d_min), L.vp(d_max), m, C.byref(pp), C.byref(cam), C.c_float(bounds[0]),
                                      C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), C.c_float(log_scale_factor), nlevels,
                                      C.c_float(cos_limit), L.vp(out["x1"]), L.vp(out["y1"]), opt("x1r"), L.vp(out["x2"]), L.vp(out["y2"]), opt("x2r"),
                                      L.vp(out["level"]), L.vp(out["view_cos"]), L.vp(out["in_view"]), device, _stream(stream)), "plf_frustum_lines")
