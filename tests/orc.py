"""ctypes bindings of the CPU oracle (oracle/liborc.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                     ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                     ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"), ("sPointInOctaveY", "<f4"),
                     ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"), ("lineLength", "<f4"), ("numOfPixels", "<i4")])
assert KP_DTYPE.itemsize == 28 and KL_DTYPE.itemsize == 68


class OrbDebug(C.Structure):
    _fields_ = [("want_planes", C.c_int), ("ncand", C.c_int * 16), ("nsel", C.c_int * 16), ("ncells", C.c_int * 16),
                ("lw", C.c_int * 16), ("lh", C.c_int * 16), ("pyr", C.c_void_p * 16), ("blur", C.c_void_p * 16),
                ("level_kps", C.c_void_p * 16)]


class LsdDebug(C.Structure):
    _fields_ = [("want_maps", C.c_int), ("sw", C.c_int), ("sh", C.c_int), ("nregions", C.c_int), ("min_reg_size", C.c_int),
                ("max_grad", C.c_double), ("scaled", C.c_void_p), ("angles", C.c_void_p), ("modgrad", C.c_void_p),
                ("used", C.c_void_p)]


class Frame(C.Structure):
    _fields_ = [("n", C.c_int), ("ux", C.c_void_p), ("uy", C.c_void_p), ("octave", C.c_void_p), ("uright", C.c_void_p),
                ("desc", C.c_void_p), ("angle", C.c_void_p), ("minx", C.c_float), ("miny", C.c_float), ("maxx", C.c_float),
                ("maxy", C.c_float), ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("scale_factors", C.c_void_p),
                ("nlevels", C.c_int)]


class MapPoints(C.Structure):
    _fields_ = [("m", C.c_int), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p),
                ("level", C.c_void_p), ("view_cos", C.c_void_p), ("in_view", C.c_void_p), ("desc", C.c_void_p),
                ("obs_positive", C.c_void_p)]


class LastFrame(C.Structure):
    _fields_ = [("n", C.c_int), ("has_mp", C.c_void_p), ("outlier", C.c_void_p), ("xw", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("mp_desc", C.c_void_p), ("obs_positive", C.c_void_p)]


class Points3D(C.Structure):
    _fields_ = [("m", C.c_int), ("xw", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p), ("max_dist", C.c_void_p),
                ("desc", C.c_void_p), ("valid", C.c_void_p)]


class KfPose(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float), ("log_scale_factor", C.c_float), ("inv_level_sigma2", C.c_void_p)]


class LineFrame(C.Structure):
    _fields_ = [("n", C.c_int), ("pt_x", C.c_void_p), ("pt_y", C.c_void_p), ("angle", C.c_void_p), ("octave", C.c_void_p),
                ("desc", C.c_void_p), ("scale_factors", C.c_void_p)]


class MapLines(C.Structure):
    _fields_ = [("m", C.c_int), ("x1", C.c_void_p), ("y1", C.c_void_p), ("x2", C.c_void_p), ("y2", C.c_void_p),
                ("level", C.c_void_p), ("view_cos", C.c_void_p), ("in_view", C.c_void_p), ("desc", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_radius_by_viewing_cos.restype = C.c_float
        _lib.orc_radius_by_viewing_cos.argtypes = [C.c_float]
        _lib.orc_ic_angle.restype = C.c_float
        _lib.orc_free.argtypes = [C.c_void_p]
    return _lib


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def orb_tables(nfeatures, scale_factor, nlevels):
    L = lib()
    sc = np.zeros(16, np.float32); inv = np.zeros(16, np.float32); s2 = np.zeros(16, np.float32); is2 = np.zeros(16, np.float32)
    per = np.zeros(16, np.int32); umax = np.zeros(16, np.int32)
    L.orc_orb_tables(C.c_int(nfeatures), C.c_float(scale_factor), C.c_int(nlevels), p(sc), p(inv), p(s2), p(is2), p(per), p(umax))
    return dict(scale=sc[:nlevels], inv=inv[:nlevels], sigma2=s2[:nlevels], invsigma2=is2[:nlevels], perLevel=per[:nlevels], umax=umax)


def distribute_octree(kps, minX, maxX, minY, maxY, N):
    L = lib()
    kin = np.zeros(len(kps), KP_DTYPE)
    kin["x"] = kps[:, 0]; kin["y"] = kps[:, 1]; kin["response"] = kps[:, 2]; kin["size"] = 7; kin["angle"] = -1
    kin["class_id"] = np.arange(len(kps))
    out = np.zeros(len(kps) + 8, KP_DTYPE)
    n = L.orc_distribute_octree(p(kin), C.c_int(len(kps)), C.c_int(minX), C.c_int(maxX), C.c_int(minY), C.c_int(maxY),
                                C.c_int(N), p(out), C.c_int(len(out)))
    return out[:n]


def orb_extract(gray, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, debug=False):
    L = lib()
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = nfeatures * 2 + 512
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    dbg = OrbDebug(); dbg.want_planes = 1 if debug else 0
    n = L.orc_orb_extract(p(gray), C.c_int(w), C.c_int(h), C.c_ssize_t(w), C.c_int(nfeatures), C.c_float(scale_factor),
                          C.c_int(nlevels), C.c_int(ini_th), C.c_int(min_th), p(kps), p(desc), C.c_int(cap), C.byref(dbg))
    assert 0 <= n <= cap
    res = dict(kps=kps[:n].copy(), desc=desc[:n].copy(), ncand=list(dbg.ncand[:nlevels]), nsel=list(dbg.nsel[:nlevels]),
               ncells=list(dbg.ncells[:nlevels]), lw=list(dbg.lw[:nlevels]), lh=list(dbg.lh[:nlevels]))
    if debug:
        res["pyr"], res["blur"], res["level_kps"] = [], [], []
        for l in range(nlevels):
            lw, lh = dbg.lw[l], dbg.lh[l]
            buf = (C.c_uint8 * ((lw + 38) * (lh + 38))).from_address(dbg.pyr[l])
            res["pyr"].append(np.frombuffer(buf, np.uint8).reshape(lh + 38, lw + 38).copy())
            buf = (C.c_uint8 * (lw * lh)).from_address(dbg.blur[l])
            res["blur"].append(np.frombuffer(buf, np.uint8).reshape(lh, lw).copy())
            ns = dbg.nsel[l]
            if ns > 0:
                buf = (C.c_uint8 * (28 * ns)).from_address(dbg.level_kps[l])
                res["level_kps"].append(np.frombuffer(buf, KP_DTYPE).copy())
            else:
                res["level_kps"].append(np.zeros(0, KP_DTYPE))
            L.orc_free(dbg.pyr[l]); L.orc_free(dbg.blur[l]); L.orc_free(dbg.level_kps[l])
    return res


def lsd_detect(gray, seed_order=0, maps=False):
    L = lib()
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = 1 << 15
    lines = np.zeros((cap, 4), np.float32)
    dbg = LsdDebug(); dbg.want_maps = 1 if maps else 0
    n = L.orc_lsd_detect(p(gray), C.c_int(w), C.c_int(h), C.c_ssize_t(w), C.c_int(seed_order), p(lines), C.c_int(cap), C.byref(dbg))
    res = dict(lines=lines[:n].copy(), sw=dbg.sw, sh=dbg.sh, nregions=dbg.nregions, min_reg_size=dbg.min_reg_size, max_grad=dbg.max_grad)
    if maps:
        npx = dbg.sw * dbg.sh
        for name, ct, dt in (("scaled", C.c_double, np.float64), ("angles", C.c_double, np.float64), ("modgrad", C.c_double, np.float64), ("used", C.c_uint8, np.uint8)):
            addr = getattr(dbg, name)
            res[name] = np.frombuffer((ct * npx).from_address(addr), dt).reshape(dbg.sh, dbg.sw).copy()
            L.orc_free(addr)
    return res


LBD_BLURRED, LBD_RAW = 0, 1   # what BinaryDescriptor's Sobel reads (plf_line_params.lbd_sobel_input)


def line_extract(gray, nkeep=100, seed_order=0, lbd_sobel_input=LBD_BLURRED):
    L = lib()
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = nkeep
    kl = np.zeros(cap, KL_DTYPE); desc = np.zeros((cap, 32), np.uint8); eq = np.zeros((cap, 3), np.float64)
    nd = C.c_int(0)
    n = L.orc_line_extract_ex(p(gray), C.c_int(w), C.c_int(h), C.c_ssize_t(w), C.c_int(nkeep), C.c_int(seed_order), p(kl), p(desc),
                              p(eq), C.c_int(cap), C.byref(nd), C.c_int(lbd_sobel_input))
    return dict(kl=kl[:n].copy(), desc=desc[:n].copy(), eq=eq[:n].copy(), ndetected=nd.value)


# ---------------------------------------------------------------- matcher wrappers
def _frame(kps, desc, uright, scale, bounds):
    ux = np.ascontiguousarray(kps["x"]); uy = np.ascontiguousarray(kps["y"]); oc = np.ascontiguousarray(kps["octave"])
    ang = np.ascontiguousarray(kps["angle"])
    ur = np.ascontiguousarray(uright if uright is not None else np.full(len(kps), -1, np.float32), np.float32)
    desc = np.ascontiguousarray(desc); scale = np.ascontiguousarray(scale, np.float32)
    F = Frame()
    F.n = len(kps); F.ux = p(ux).value; F.uy = p(uy).value; F.octave = p(oc).value; F.uright = p(ur).value; F.desc = p(desc).value
    F.angle = p(ang).value
    F.minx, F.miny, F.maxx, F.maxy = bounds
    F.grid_inv_w = np.float32(64.0) / np.float32(bounds[2] - bounds[0]); F.grid_inv_h = np.float32(48.0) / np.float32(bounds[3] - bounds[1])
    F.scale_factors = p(scale).value; F.nlevels = len(scale)
    return F, (ux, uy, oc, ur, desc, scale, ang)


def search_by_projection_map(kps, desc, uright, scale, bounds, mp, th, nnratio, match_init):
    L = lib()
    F, keep = _frame(kps, desc, uright, scale, bounds)
    arrs = {k: np.ascontiguousarray(v) for k, v in mp.items()}
    M = MapPoints()
    M.m = len(arrs["desc"])
    for k in ("proj_x", "proj_y", "proj_xr", "level", "view_cos", "in_view", "desc"):
        setattr(M, k, p(arrs[k]).value)
    M.obs_positive = p(arrs["obs_positive"]).value if "obs_positive" in arrs else None
    match = np.ascontiguousarray(match_init, np.int32).copy()
    n = L.orc_search_by_projection_map(C.byref(F), C.byref(M), C.c_float(th), C.c_float(nnratio), p(match))
    return match, n


def search_by_projection_last(kps, desc, uright, scale, bounds, last, pose, th, mono, check_ori, match_init):
    L = lib()
    F, keep = _frame(kps, desc, uright, scale, bounds)
    lk = last["keys"]
    oc = np.ascontiguousarray(lk["octave"]); ang = np.ascontiguousarray(lk["angle"])
    arrs = {k: np.ascontiguousarray(last[k]) for k in ("has_mappoint", "outlier", "world_pos", "mp_desc")}
    Lf = LastFrame()
    Lf.n = len(lk); Lf.has_mp = p(arrs["has_mappoint"]).value; Lf.outlier = p(arrs["outlier"]).value; Lf.xw = p(arrs["world_pos"]).value
    Lf.octave = p(oc).value; Lf.angle = p(ang).value; Lf.mp_desc = p(arrs["mp_desc"]).value
    obs = np.ascontiguousarray(last["obs_positive"], np.uint8) if last.get("obs_positive") is not None else None
    Lf.obs_positive = p(obs).value if obs is not None else None
    R = {k: np.ascontiguousarray(np.asarray(pose[k], np.float32).ravel()) for k in ("Rcw", "tcw", "Rlw", "tlw")}
    match = np.ascontiguousarray(match_init, np.int32).copy()
    n = L.orc_search_by_projection_last(C.byref(F), C.byref(Lf), p(R["Rcw"]), p(R["tcw"]), p(R["Rlw"]), p(R["tlw"]), C.c_float(pose["fx"]),
                                        C.c_float(pose["fy"]), C.c_float(pose["cx"]), C.c_float(pose["cy"]), C.c_float(pose["bf"]),
                                        C.c_float(pose["b"]), C.c_float(th), C.c_int(mono), C.c_int(check_ori), p(match))
    return match, n


def knn2(q, t):
    L = lib()
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.int32)
    L.orc_knn2_hamming(p(q), C.c_int(len(q)), p(t), C.c_int(len(t)), p(idx), p(dist))
    return idx, dist


def line_mad(dist):
    """Frame::lineDescriptorMAD on an (n, 2) int32 distance table: (nn_mad, nn12_mad)"""
    L = lib()
    dist = np.ascontiguousarray(dist, np.int32)
    a = C.c_double(0); b = C.c_double(0)
    L.orc_line_mad(p(dist), C.c_int(len(dist)), C.byref(a), C.byref(b))
    return a.value, b.value


def line_segment_overlap(a, b, c, d):
    L = lib()
    L.orc_line_segment_overlap.restype = C.c_double
    return L.orc_line_segment_overlap(C.c_double(a), C.c_double(b), C.c_double(c), C.c_double(d))


def lines_search_for_triangulation(desc1, desc2, has_ml1, has_ml2, stereo1, stereo2, only_stereo, mad_factor=0.1):
    L = lib()
    a = [np.ascontiguousarray(x, np.uint8) for x in (desc1, desc2, has_ml1, has_ml2, stereo1, stereo2)]
    match = np.full(len(desc1), -1, np.int32)
    n = L.orc_lines_search_for_triangulation(p(a[0]), C.c_int(len(desc1)), p(a[1]), C.c_int(len(desc2)), p(a[2]), p(a[3]), p(a[4]), p(a[5]),
                                             C.c_int(int(only_stereo)), C.c_double(mad_factor), p(match))
    return match, n


def lines_fuse(kf_desc, ml_desc, valid):
    L = lib()
    a = [np.ascontiguousarray(x, np.uint8) for x in (kf_desc, ml_desc, valid)]
    best = np.full(len(ml_desc), -1, np.int32)
    n = L.orc_lines_fuse(p(a[0]), C.c_int(len(kf_desc)), p(a[1]), p(a[2]), C.c_int(len(ml_desc)), p(best))
    return best, n


def match_lines_knn(last_desc, cur_desc, has_ml):
    L = lib()
    a = np.ascontiguousarray(last_desc); b = np.ascontiguousarray(cur_desc); hm = np.ascontiguousarray(has_ml, np.uint8)
    match = np.full(len(b), -1, np.int32)
    n = L.orc_match_lines_knn(p(a), C.c_int(len(a)), p(b), C.c_int(len(b)), p(hm), p(match))
    return match, n


def search_lines_by_projection(kl, ldesc, scale, ml, th, nnratio, match_init):
    L = lib()
    px = np.ascontiguousarray(kl["pt_x"]); py = np.ascontiguousarray(kl["pt_y"]); an = np.ascontiguousarray(kl["angle"])
    oc = np.ascontiguousarray(kl["octave"]); d = np.ascontiguousarray(ldesc); sc = np.ascontiguousarray(scale, np.float32)
    F = LineFrame()
    F.n = len(kl); F.pt_x = p(px).value; F.pt_y = p(py).value; F.angle = p(an).value; F.octave = p(oc).value; F.desc = p(d).value
    F.scale_factors = p(sc).value
    arrs = {k: np.ascontiguousarray(v) for k, v in ml.items()}
    M = MapLines()
    M.m = len(arrs["desc"])
    for k in ("x1", "y1", "x2", "y2", "level", "view_cos", "in_view", "desc"):
        setattr(M, k, p(arrs[k]).value)
    match = np.ascontiguousarray(match_init, np.int32).copy()
    n = L.orc_search_by_projection_lines(C.byref(F), C.byref(M), C.c_float(th), C.c_float(nnratio), p(match))
    return match, n


# ---------------------------------------------------------------- frame tail / frustum wrappers
def rgb_to_gray(rgb, bgr=True):
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb); out = np.zeros((h, w), np.uint8)
    lib().orc_rgb_to_gray(p(rgb), C.c_int(w), C.c_int(h), C.c_ssize_t(3 * w), C.c_int(int(bgr)), p(out), C.c_ssize_t(w))
    return out


def depth_to_float(d16, factor):
    h, w = d16.shape
    d16 = np.ascontiguousarray(d16); out = np.zeros((h, w), np.float32)
    lib().orc_depth_to_float(p(d16), C.c_int(w), C.c_int(h), C.c_ssize_t(w), C.c_float(factor), p(out))
    return out


def frame_tail(kps, depth, cam9, bf):
    kps = np.ascontiguousarray(kps); un = np.zeros(len(kps), KP_DTYPE)
    cam = np.ascontiguousarray(cam9, np.float32)
    lib().orc_undistort_keypoints(p(kps), C.c_int(len(kps)), p(cam), p(un))
    ur = np.zeros(len(kps), np.float32); kd = np.zeros(len(kps), np.float32)
    h, w = depth.shape
    depth = np.ascontiguousarray(depth, np.float32)
    lib().orc_stereo_from_rgbd(p(kps), p(un), C.c_int(len(kps)), p(depth), C.c_int(w), C.c_int(h), C.c_float(bf), p(ur), p(kd))
    return un, ur, kd


def line_tail(kls, depth, cam9, bf):
    kls = np.ascontiguousarray(kls); n = len(kls); un = np.zeros(n, KL_DTYPE)
    cam = np.ascontiguousarray(cam9, np.float32)
    out = [np.zeros(n, np.float32) for _ in range(4)]
    if depth is not None:
        h, w = depth.shape
        depth = np.ascontiguousarray(depth, np.float32)
    else:
        h = w = 0
    lib().orc_line_tail(p(kls), C.c_int(n), p(cam), p(depth) if depth is not None else None, C.c_int(w), C.c_int(h), C.c_float(bf), p(un),
                        p(out[0]), p(out[1]), p(out[2]), p(out[3]))
    return un, out[0], out[1], out[2], out[3]


def is_in_frustum(xw, normal, dmin, dmax, Rcw, tcw, Ow, cam4, bounds, bf, logsf, nlevels, coslim):
    m = len(xw)
    a = [np.ascontiguousarray(v, np.float32) for v in (xw, normal, dmin, dmax, np.asarray(Rcw).ravel(), tcw, Ow, cam4, bounds)]
    px = np.zeros(m, np.float32); py = np.zeros(m, np.float32); pxr = np.zeros(m, np.float32); lv = np.zeros(m, np.int32)
    vc = np.zeros(m, np.float32); iv = np.zeros(m, np.uint8)
    lib().orc_is_in_frustum(p(a[0]), p(a[1]), p(a[2]), p(a[3]), C.c_int(m), p(a[4]), p(a[5]), p(a[6]), p(a[7]), p(a[8]), C.c_float(bf),
                            C.c_float(logsf), C.c_int(nlevels), C.c_float(coslim), p(px), p(py), p(pxr), p(lv), p(vc), p(iv))
    return dict(proj_x=px, proj_y=py, proj_xr=pxr, level=lv, view_cos=vc, in_view=iv)


def is_in_frustum_line(xw6, normal, dmin, dmax, Rcw, tcw, Ow, cam4, bounds, bf, logsf, nlevels, coslim):
    m = len(xw6)
    a = [np.ascontiguousarray(v, np.float32) for v in (xw6, normal, dmin, dmax, np.asarray(Rcw).ravel(), tcw, Ow, cam4, bounds)]
    o6 = np.zeros((m, 6), np.float32); lv = np.zeros(m, np.int32); vc = np.zeros(m, np.float32); iv = np.zeros(m, np.uint8)
    lib().orc_is_in_frustum_line(p(a[0]), p(a[1]), p(a[2]), p(a[3]), C.c_int(m), p(a[4]), p(a[5]), p(a[6]), p(a[7]), p(a[8]), C.c_float(bf),
                                 C.c_float(logsf), C.c_int(nlevels), C.c_float(coslim), p(o6), p(lv), p(vc), p(iv))
    return dict(x1=o6[:, 0].copy(), y1=o6[:, 1].copy(), x1r=o6[:, 2].copy(), x2=o6[:, 3].copy(), y2=o6[:, 4].copy(), x2r=o6[:, 5].copy(),
                level=lv, view_cos=vc, in_view=iv)


def search_by_bow(kf_desc, f_desc, kf_angle, f_angle, kf_has_mp, kf_nodes, f_nodes, nnratio, check_ori):
    """kf_nodes / f_nodes = (node_id uint32[K], node_start int32[K+1], feat int32[*]) flattened DBoW2 feature vectors"""
    L = lib()
    a = [np.ascontiguousarray(x) for x in (kf_desc, f_desc, np.asarray(kf_angle, np.float32), np.asarray(f_angle, np.float32), np.asarray(kf_has_mp, np.uint8))]
    kn = [np.ascontiguousarray(kf_nodes[0], np.uint32), np.ascontiguousarray(kf_nodes[1], np.int32), np.ascontiguousarray(kf_nodes[2], np.int32)]
    fn = [np.ascontiguousarray(f_nodes[0], np.uint32), np.ascontiguousarray(f_nodes[1], np.int32), np.ascontiguousarray(f_nodes[2], np.int32)]
    match = np.full(len(a[1]), -1, np.int32)
    n = L.orc_search_by_bow(C.c_int(len(a[0])), C.c_int(len(a[1])), p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(a[4]), C.c_int(len(kn[0])), p(kn[0]), p(kn[1]),
                            p(kn[2]), C.c_int(len(fn[0])), p(fn[0]), p(fn[1]), p(fn[2]), C.c_float(nnratio), C.c_int(int(check_ori)), p(match))
    return match, n


def search_by_bow_kf(desc1, desc2, angle1, angle2, has_mp1, has_mp2, nodes1, nodes2, nnratio, check_ori):
    L = lib()
    a = [np.ascontiguousarray(x) for x in (desc1, desc2, np.asarray(angle1, np.float32), np.asarray(angle2, np.float32), np.asarray(has_mp1, np.uint8),
                                           np.asarray(has_mp2, np.uint8))]
    n1 = [np.ascontiguousarray(nodes1[0], np.uint32), np.ascontiguousarray(nodes1[1], np.int32), np.ascontiguousarray(nodes1[2], np.int32)]
    n2 = [np.ascontiguousarray(nodes2[0], np.uint32), np.ascontiguousarray(nodes2[1], np.int32), np.ascontiguousarray(nodes2[2], np.int32)]
    match = np.full(len(a[0]), -1, np.int32)
    n = L.orc_search_by_bow_kf(C.c_int(len(a[0])), C.c_int(len(a[1])), p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(a[4]), p(a[5]), C.c_int(len(n1[0])), p(n1[0]),
                               p(n1[1]), p(n1[2]), C.c_int(len(n2[0])), p(n2[0]), p(n2[1]), p(n2[2]), C.c_float(nnratio), C.c_int(int(check_ori)), p(match))
    return match, n


def search_by_projection_reloc(kps, desc, scale, bounds, kf, pose, log_scale_factor, th, orbdist, check_ori, match_init):
    """kf: dict(keys (angle), valid, world_pos, min_dist, max_dist, mp_desc); pose: Rcw, tcw, fx, fy, cx, cy"""
    L = lib()
    F, keep = _frame(kps, desc, None, scale, bounds)
    ang = np.ascontiguousarray(kf["keys"]["angle"]); oc = np.ascontiguousarray(kf["keys"]["octave"])
    arrs = {k: np.ascontiguousarray(kf[k]) for k in ("valid", "world_pos", "mp_desc")}
    mn = np.ascontiguousarray(kf["min_dist"], np.float32); mx = np.ascontiguousarray(kf["max_dist"], np.float32)
    Lf = LastFrame()
    Lf.n = len(ang); Lf.has_mp = p(arrs["valid"]).value; Lf.outlier = None; Lf.xw = p(arrs["world_pos"]).value
    Lf.octave = p(oc).value; Lf.angle = p(ang).value; Lf.mp_desc = p(arrs["mp_desc"]).value
    R = {k: np.ascontiguousarray(np.asarray(pose[k], np.float32).ravel()) for k in ("Rcw", "tcw")}
    match = np.ascontiguousarray(match_init, np.int32).copy()
    n = L.orc_search_by_projection_reloc(C.byref(F), C.byref(Lf), p(mn), p(mx), p(R["Rcw"]), p(R["tcw"]), C.c_float(pose["fx"]), C.c_float(pose["fy"]),
                                         C.c_float(pose["cx"]), C.c_float(pose["cy"]), C.c_float(log_scale_factor), C.c_float(th), C.c_int(orbdist),
                                         C.c_int(check_ori), p(match))
    return match, n



def _points3d(pts):
    arrs = dict(xw=np.ascontiguousarray(pts["xw"], np.float32), normal=np.ascontiguousarray(pts["normal"], np.float32),
                min_dist=np.ascontiguousarray(pts["min_dist"], np.float32), max_dist=np.ascontiguousarray(pts["max_dist"], np.float32),
                desc=np.ascontiguousarray(pts["desc"], np.uint8), valid=np.ascontiguousarray(pts["valid"], np.uint8))
    P = Points3D()
    P.m = len(arrs["valid"])
    for k, v in arrs.items():
        setattr(P, k, p(v).value)
    return P, arrs


def _kf_pose(pose):
    Cp = KfPose()
    Cp.Rcw[:] = [float(x) for x in np.asarray(pose["Rcw"], np.float32).ravel()]
    Cp.tcw[:] = [float(x) for x in np.asarray(pose["tcw"], np.float32).ravel()]
    Cp.Ow[:] = [float(x) for x in np.asarray(pose["Ow"], np.float32).ravel()]
    for k in ("fx", "fy", "cx", "cy", "bf", "log_scale_factor"):
        setattr(Cp, k, float(pose[k]))
    isg = np.ascontiguousarray(pose["inv_sigma2"], np.float32)
    Cp.inv_level_sigma2 = p(isg).value
    return Cp, isg


def fuse(kps, desc, uright, scale, bounds, pose, pts, th):
    """ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th): (best_idx, best_dist, nFused)"""
    L = lib()
    F, keep = _frame(kps, desc, uright, scale, bounds)
    P, keep2 = _points3d(pts)
    Cp, keep3 = _kf_pose(pose)
    best = np.zeros(P.m, np.int32); bd = np.zeros(P.m, np.int32)
    L.orc_fuse.restype = C.c_int
    n = L.orc_fuse(C.byref(F), C.byref(Cp), C.byref(P), C.c_float(th), p(best), p(bd))
    return best, bd, n


def sim3_decompose(Scw):
    L = lib()
    S = np.ascontiguousarray(Scw, np.float32)
    R = np.zeros(9, np.float32); t = np.zeros(3, np.float32); Ow = np.zeros(3, np.float32)
    L.orc_sim3_decompose(p(S), p(R), p(t), p(Ow))
    return R.reshape(3, 3), t, Ow


def _sim3_pose(Scw, intr, nlevels):
    R, t, Ow = sim3_decompose(Scw)
    pose = dict(Rcw=R, tcw=t, Ow=Ow, inv_sigma2=np.ones(nlevels, np.float32), **intr)
    return _kf_pose(pose)


def fuse_sim3(kps, desc, scale, bounds, Scw, intr, pts, th):
    """ORBmatcher::Fuse(KeyFrame*, Scw, points, th, vpReplacePoint): (best_idx, nFused)"""
    L = lib()
    F, keep = _frame(kps, desc, None, scale, bounds)
    P, keep2 = _points3d(pts)
    Cp, keep3 = _sim3_pose(Scw, intr, len(scale))
    best = np.zeros(P.m, np.int32)
    n = L.orc_fuse_sim3(C.byref(F), C.byref(Cp), C.byref(P), C.c_float(th), p(best))
    return best, n


def search_by_projection_sim3(kps, desc, scale, bounds, Scw, intr, pts, th, match_init):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, points, vpMatched, th): (match_of_kp, nmatches)"""
    L = lib()
    F, keep = _frame(kps, desc, None, scale, bounds)
    P, keep2 = _points3d(pts)
    Cp, keep3 = _sim3_pose(Scw, intr, len(scale))
    match = np.ascontiguousarray(match_init, np.int32).copy()
    n = L.orc_search_by_projection_sim3(C.byref(F), C.byref(Cp), C.byref(P), C.c_int(int(th)), p(match))
    return match, n


def search_by_sim3(c, bounds=(0.0, 0.0, 640.0, 480.0)):
    """ORBmatcher::SearchBySim3 on a case dict of refgen.load_sim3_cases: (match12, nFound)"""
    L = lib()
    F1, k1 = _frame(c["kps1"], c["desc1"], None, c["scale"], bounds)
    F2, k2 = _frame(c["kps2"], c["desc2"], None, c["scale"], bounds)
    P1, k3 = _points3d(c["pts1"]); P2, k4 = _points3d(c["pts2"])
    C1, k5 = _kf_pose(c["pose1"]); C2, k6 = _kf_pose(c["pose2"])
    R12 = np.ascontiguousarray(c["R12"], np.float32); t12 = np.ascontiguousarray(c["t12"], np.float32)
    match = np.zeros(c["n"], np.int32)
    n = L.orc_search_by_sim3(C.byref(F1), C.byref(F2), C.byref(C1), C.byref(C2), C.c_float(c["s12"]), p(R12), p(t12), C.c_float(c["th"]),
                             C.byref(P1), C.byref(P2), p(match))
    return match, n


def epipole(Cw, R2w, t2w, fx, fy, cx, cy):
    L = lib()
    a = [np.ascontiguousarray(x, np.float32) for x in (Cw, np.asarray(R2w).ravel(), t2w)]
    ex = C.c_float(); ey = C.c_float()
    L.orc_epipole(p(a[0]), p(a[1]), p(a[2]), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.byref(ex), C.byref(ey))
    return ex.value, ey.value


def search_for_triangulation(c):
    """ORBmatcher::SearchForTriangulation on a case dict of refgen.load_triangulation_cases: (match12, nmatches)"""
    L = lib()
    ex, ey = epipole(c["Ow1"], c["R2w"], c["t2w"], c["fx"], c["fy"], c["cx"], c["cy"])
    k1, k2 = c["kps1"], c["kps2"]
    arr = dict(x1=k1["x"], y1=k1["y"], a1=k1["angle"], x2=k2["x"], y2=k2["y"], a2=k2["angle"], o2=k2["octave"])
    arr = {k: np.ascontiguousarray(v) for k, v in arr.items()}
    F = np.ascontiguousarray(c["F12"], np.float32)
    n1 = c["nodes1"]; n2 = c["nodes2"]
    match = np.zeros(c["n"], np.int32)
    n = L.orc_search_for_triangulation(c["n"], c["n"], p(arr["x1"]), p(arr["y1"]), p(arr["a1"]), p(c["uright1"]), p(c["desc1"]), p(c["has_mp1"]),
                                       p(arr["x2"]), p(arr["y2"]), p(arr["a2"]), p(arr["o2"]), p(c["uright2"]), p(c["desc2"]), p(c["has_mp2"]),
                                       len(n1[0]), p(n1[0]), p(n1[1]), p(n1[2]), len(n2[0]), p(n2[0]), p(n2[1]), p(n2[2]), p(F), C.c_float(ex), C.c_float(ey),
                                       p(c["scale"]), p(c["sigma2"]), int(c["only_stereo"]), int(c["check"]), p(match))
    return match, n


def assign_grid(x, y, bounds):
    """Frame::AssignFeaturesToGrid as restated in match_oracle.c: (cell_start, cell_idx)"""
    L = lib()
    n = len(x)
    kps = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("octave", "<i4"), ("angle", "<f4")])
    kps["x"] = x; kps["y"] = y
    F, keep = _frame(kps, np.zeros((max(n, 1), 32), np.uint8), None, np.ones(1, np.float32), bounds)
    cs = np.zeros(64 * 48 + 1, np.int32); ci = np.zeros(max(n, 1), np.int32)
    L.orc_assign_grid(C.byref(F), p(cs), p(ci))
    return cs, ci[:cs[-1]]
