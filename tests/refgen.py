"""Deterministic input generators shared with oracle/refprobe/probe.cpp (same 64-bit LCG), so that
the golden fixtures under tests/golden/ only need to store OUTPUTS of the reference binary."""
import numpy as np

_A = 6364136223846793005
_C = 1442695040888963407
_M = (1 << 64) - 1


class LCG:
    def __init__(self, seed):
        self.s = seed & _M

    def u32(self):
        self.s = (self.s * _A + _C) & _M
        return self.s >> 32

    def below(self, n):
        return (self.u32() * n) >> 32

    def vec_u32(self, count):
        """next `count` outputs, vectorised (uint64 wrap-around arithmetic)."""
        if count == 0:
            return np.zeros(0, np.uint64)
        with np.errstate(over="ignore"):
            a = np.full(count, _A, np.uint64)
            A = np.multiply.accumulate(a)                      # a^(i+1)
            Ash = np.concatenate(([np.uint64(1)], A[:-1]))     # a^i
            S = np.add.accumulate(Ash)                         # 1 + a + ... + a^i
            st = A * np.uint64(self.s) + S * np.uint64(_C)
        self.s = int(st[-1])
        return st >> np.uint64(32)

    def vec_below(self, n, count):
        return (self.vec_u32(count) * np.uint64(n)) >> np.uint64(32)


def synth_image(seed, w, h):
    """mirror of probe.cpp:synth_image"""
    g = LCG(seed)
    base = 96 + g.below(64)
    img = np.full((h, w), base, np.int32)
    nrect = 40 + g.below(40)
    for _ in range(nrect):
        x0 = g.below(w); y0 = g.below(h)
        rw = 8 + g.below(w // 4); rh = 8 + g.below(h // 4)
        v = g.below(256)
        img[y0:min(y0 + rh, h), x0:min(x0 + rw, w)] = v
    noise = g.vec_below(9, w * h).astype(np.int32).reshape(h, w) - 4
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def octree_case(seed, cluster, resp_levels):
    """mirror of the DistributeOctTree case generator in probe.cpp. Returns (W,H,N,kps[nk,3]=x,y,response)"""
    g = LCG(seed)
    W = 64 + g.below(1200); H = 48 + g.below(900)
    if W < H:
        W, H = H, W
    c = seed - 1000
    nk = 1 + g.below(60 if c % 4 == 0 else 3500)
    N = 1 + g.below(450)
    k = np.zeros((nk, 3), np.float32)
    for i in range(nk):
        if cluster:
            x = g.below(W // 4 + 1) + g.below(2) * (W // 2); y = g.below(H // 3 + 1)
        else:
            x = g.below(W); y = g.below(H)
        if x >= W:
            x = W - 1
        k[i] = (x, y, 7 + g.below(resp_levels))
    return W, H, N, k


def load_search_map_cases(path):
    """tests/golden/ref_glue_search_map.json -> list of dicts with numpy inputs and the reference binary's result"""
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        N, M = c["n"], c["m"]
        kps = np.zeros(N, dtype=kp_dtype)
        kps["x"] = f32(c["x"]); kps["y"] = f32(c["y"]); kps["octave"] = np.array(c["octave"], np.int32)
        mp = dict(proj_x=f32(c["proj_x"]), proj_y=f32(c["proj_y"]), proj_xr=f32(c["proj_xr"]), level=np.array(c["level"], np.int32),
                  view_cos=f32(c["view_cos"]), in_view=np.array(c["in_view"], np.uint8),
                  desc=np.frombuffer(bytes.fromhex(c["mp_desc"]), np.uint8).reshape(M, 32).copy(), obs_positive=np.array(c["obs_positive"], np.uint8))
        out.append(dict(kps=kps, desc=np.frombuffer(bytes.fromhex(c["desc"]), np.uint8).reshape(N, 32).copy(),
                        uright=f32(c["uright"]) if c["with_uright"] else None, scale=f32(c["scale"]), mp=mp,
                        th=float(np.array([c["th_bits"]], np.uint32).view(np.float32)[0]), init=np.array(c["init"], np.int32),
                        match=np.array(c["match"], np.int32), nmatches=c["nmatches"]))
    return out


def load_bow_cases(path):
    """tests/golden/ref_glue_bow.json -> list of dicts with numpy inputs and the reference binary's result"""
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    out = []
    for c in json.load(open(path))["cases"]:
        out.append(dict(kf_desc=np.frombuffer(bytes.fromhex(c["kf_desc"]), np.uint8).reshape(c["n_kf"], 32).copy(),
                        f_desc=np.frombuffer(bytes.fromhex(c["f_desc"]), np.uint8).reshape(c["n_f"], 32).copy(),
                        kf_angle=f32(c["kf_angle"]), f_angle=f32(c["f_angle"]), kf_has_mp=np.array(c["kf_has_mp"], np.uint8),
                        kf_nodes=(np.array(c["kf_node_id"], np.uint32), np.array(c["kf_node_start"], np.int32), np.array(c["kf_feat"], np.int32)),
                        f_nodes=(np.array(c["f_node_id"], np.uint32), np.array(c["f_node_start"], np.int32), np.array(c["f_feat"], np.int32)),
                        nnratio=float(c["nnratio"]), check=int(c["check_orientation"]), match=np.array(c["match"], np.int32), nmatches=c["nmatches"]))
    return out



def load_search_last_cases(path):
    """tests/golden/ref_glue_search_last.json -> inputs of SearchByProjection(CurrentFrame, LastFrame) + the binary's result"""
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        NC, NL = c["n_cur"], c["n_last"]
        kps = np.zeros(NC, dtype=kp_dtype)
        kps["x"] = f32(c["x"]); kps["y"] = f32(c["y"]); kps["angle"] = f32(c["angle"]); kps["octave"] = np.array(c["octave"], np.int32)
        lk = np.zeros(NL, dtype=kp_dtype)
        lk["angle"] = f32(c["last_angle"]); lk["octave"] = np.array(c["last_octave"], np.int32)
        cam = f32(c["cam"]); Tc = f32(c["Tcw"]).reshape(4, 4); Tl = f32(c["Tlw"]).reshape(4, 4)
        last = dict(keys=lk, has_mappoint=np.array(c["last_has_mp"], np.uint8), outlier=np.array(c["last_outlier"], np.uint8),
                    world_pos=f32(c["world_pos"]).reshape(NL, 3).copy(), mp_desc=np.frombuffer(bytes.fromhex(c["mp_desc"]), np.uint8).reshape(NL, 32).copy(),
                    obs_positive=np.array(c["last_obs_positive"], np.uint8))   # Observations() > 0 of the last frame's map points (0: temporal points)
        pose = dict(Rcw=Tc[:3, :3].copy(), tcw=Tc[:3, 3].copy(), Rlw=Tl[:3, :3].copy(), tlw=Tl[:3, 3].copy(), fx=float(cam[0]), fy=float(cam[1]),
                    cx=float(cam[2]), cy=float(cam[3]), bf=float(cam[4]), b=float(cam[5]))
        match = np.array(c["match"], np.int32)
        match[match == -3] = -1     # still held by a pre-existing point without observations: free in the match protocol
        out.append(dict(kps=kps, desc=np.frombuffer(bytes.fromhex(c["desc"]), np.uint8).reshape(NC, 32).copy(), uright=f32(c["uright"]),
                        scale=f32(c["scale"]), last=last, pose=pose, th=float(cam[6]), mono=int(c["mono"]), check=int(c["check_orientation"]),
                        init=np.array(c["init"], np.int32), match=match, nmatches=c["nmatches"]))
    return out


def load_bow_kf_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    out = []
    for c in json.load(open(path))["cases"]:
        out.append(dict(desc1=np.frombuffer(bytes.fromhex(c["desc1"]), np.uint8).reshape(c["n1"], 32).copy(),
                        desc2=np.frombuffer(bytes.fromhex(c["desc2"]), np.uint8).reshape(c["n2"], 32).copy(),
                        angle1=f32(c["angle1"]), angle2=f32(c["angle2"]), has_mp1=np.array(c["has_mp1"], np.uint8), has_mp2=np.array(c["has_mp2"], np.uint8),
                        nodes1=(np.array(c["node_id1"], np.uint32), np.array(c["node_start1"], np.int32), np.array(c["feat1"], np.int32)),
                        nodes2=(np.array(c["node_id2"], np.uint32), np.array(c["node_start2"], np.int32), np.array(c["feat2"], np.int32)),
                        nnratio=float(c["nnratio"]), check=int(c["check_orientation"]), match=np.array(c["match"], np.int32), nmatches=c["nmatches"]))
    return out


def load_search_reloc_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        NC, NK = c["n_cur"], c["n_kf"]
        kps = np.zeros(NC, dtype=kp_dtype)
        kps["x"] = f32(c["x"]); kps["y"] = f32(c["y"]); kps["angle"] = f32(c["angle"]); kps["octave"] = np.array(c["octave"], np.int32)
        kk = np.zeros(NK, dtype=kp_dtype); kk["angle"] = f32(c["kf_angle"])
        cam = f32(c["cam"]); Tc = f32(c["Tcw"]).reshape(4, 4)
        kf = dict(keys=kk, valid=np.array(c["valid"], np.uint8), world_pos=f32(c["world_pos"]).reshape(NK, 3).copy(), min_dist=f32(c["min_dist"]),
                  max_dist=f32(c["max_dist"]), mp_desc=np.frombuffer(bytes.fromhex(c["mp_desc"]), np.uint8).reshape(NK, 32).copy())
        pose = dict(Rcw=Tc[:3, :3].copy(), tcw=Tc[:3, 3].copy(), fx=float(cam[0]), fy=float(cam[1]), cx=float(cam[2]), cy=float(cam[3]))
        out.append(dict(kps=kps, desc=np.frombuffer(bytes.fromhex(c["desc"]), np.uint8).reshape(NC, 32).copy(), scale=f32(c["scale"]), kf=kf, pose=pose,
                        logsf=float(cam[4]), th=float(cam[5]), orbdist=int(c["orbdist"]), check=int(c["check_orientation"]),
                        init=np.array(c["init"], np.int32), match=np.array(c["match"], np.int32), nmatches=c["nmatches"]))
    return out


def load_fuse_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        NK, M = c["n_kf"], c["m"]
        kps = np.zeros(NK, dtype=kp_dtype)
        kps["x"] = f32(c["x"]); kps["y"] = f32(c["y"]); kps["octave"] = np.array(c["octave"], np.int32)
        cam = f32(c["cam"]); Tc = f32(c["Tcw"]).reshape(4, 4)
        pose = dict(Rcw=Tc[:3, :3].copy(), tcw=Tc[:3, 3].copy(), Ow=f32(c["Ow"]), fx=float(cam[0]), fy=float(cam[1]), cx=float(cam[2]), cy=float(cam[3]),
                    bf=float(cam[4]), log_scale_factor=float(cam[5]), inv_sigma2=f32(c["inv_sigma2"]))
        pts = dict(xw=f32(c["world_pos"]).reshape(M, 3).copy(), normal=f32(c["normal"]).reshape(M, 3).copy(), min_dist=f32(c["min_dist"]),
                   max_dist=f32(c["max_dist"]), valid=np.array(c["valid"], np.uint8),
                   desc=np.frombuffer(bytes.fromhex(c["mp_desc"]), np.uint8).reshape(M, 32).copy())
        out.append(dict(kps=kps, uright=f32(c["uright"]), desc=np.frombuffer(bytes.fromhex(c["desc"]), np.uint8).reshape(NK, 32).copy(), scale=f32(c["scale"]),
                        pose=pose, pts=pts, th=float(cam[6]), best_idx=np.array(c["best_idx"], np.int32), nfused=c["nfused"]))
    return out


def load_sim3_kf_cases(path):
    """fixtures of the two Scw overloads (ref_glue_fuse_sim3.json / ref_glue_search_sim3.json)"""
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        NK, M = c["n_kf"], c["m"]
        kps = np.zeros(NK, dtype=kp_dtype)
        kps["x"] = f32(c["x"]); kps["y"] = f32(c["y"]); kps["octave"] = np.array(c["octave"], np.int32)
        cam = f32(c["cam"])
        intr = dict(fx=float(cam[0]), fy=float(cam[1]), cx=float(cam[2]), cy=float(cam[3]), bf=float(cam[4]), log_scale_factor=float(cam[5]))
        pts = dict(xw=f32(c["world_pos"]).reshape(M, 3).copy(), normal=f32(c["normal"]).reshape(M, 3).copy(), min_dist=f32(c["min_dist"]),
                   max_dist=f32(c["max_dist"]), valid=np.array(c["valid"], np.uint8),
                   desc=np.frombuffer(bytes.fromhex(c["mp_desc"]), np.uint8).reshape(M, 32).copy())
        out.append(dict(kps=kps, desc=np.frombuffer(bytes.fromhex(c["desc"]), np.uint8).reshape(NK, 32).copy(), scale=f32(c["scale"]),
                        Scw=f32(c["Scw"]).reshape(4, 4).copy(), intr=intr, pts=pts, th=float(cam[6]), init=np.array(c["init"], np.int32),
                        out=np.array(c["out"], np.int32), ret=c["ret"]))
    return out


def load_sim3_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        N = c["n"]; cam = f32(c["cam"])
        d = dict(n=N, th=float(cam[9]), s12=float(cam[10]), R12=f32(c["R12"]).reshape(3, 3).copy(), t12=f32(c["t12"]), scale=f32(c["scale"]),
                 match12=np.array(c["match12"], np.int32), nfound=c["nfound"])
        for q, sfx in enumerate(("1", "2")):
            kps = np.zeros(N, dtype=kp_dtype)
            kps["x"] = f32(c["x" + sfx]); kps["y"] = f32(c["y" + sfx]); kps["octave"] = np.array(c["octave" + sfx], np.int32)
            T = f32(c["T%sw" % sfx]).reshape(4, 4)
            d["kps" + sfx] = kps
            d["desc" + sfx] = np.frombuffer(bytes.fromhex(c["desc" + sfx]), np.uint8).reshape(N, 32).copy()
            d["pose" + sfx] = dict(Rcw=T[:3, :3].copy(), tcw=T[:3, 3].copy(), Ow=np.zeros(3, np.float32), fx=float(cam[4 * q]), fy=float(cam[4 * q + 1]),
                                   cx=float(cam[4 * q + 2]), cy=float(cam[4 * q + 3]), bf=40.0, log_scale_factor=float(cam[8]), inv_sigma2=np.ones(8, np.float32))
            d["pts" + sfx] = dict(xw=f32(c["world_pos" + sfx]).reshape(N, 3).copy(), normal=np.zeros((N, 3), np.float32), min_dist=f32(c["min_dist" + sfx]),
                                  max_dist=f32(c["max_dist" + sfx]), valid=np.array(c["valid" + sfx], np.uint8),
                                  desc=np.frombuffer(bytes.fromhex(c["mp_desc" + sfx]), np.uint8).reshape(N, 32).copy())
        out.append(d)
    return out


def load_triangulation_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = []
    for c in json.load(open(path))["cases"]:
        N = c["n"]; cam = f32(c["cam"]); T2 = f32(c["T2w"]).reshape(4, 4)
        d = dict(n=N, only_stereo=c["only_stereo"], check=c["check_orientation"], nmatches=c["nmatches"], match12=np.array(c["match12"], np.int32),
                 F12=f32(c["F12"]).reshape(3, 3).copy(), scale=f32(c["scale"]), sigma2=f32(c["sigma2"]), Ow1=f32(c["Ow1"]), R2w=T2[:3, :3].copy(), t2w=T2[:3, 3].copy(),
                 fx=float(cam[0]), fy=float(cam[1]), cx=float(cam[2]), cy=float(cam[3]))
        for sfx in ("1", "2"):
            kps = np.zeros(N, dtype=kp_dtype)
            kps["x"] = f32(c["x" + sfx]); kps["y"] = f32(c["y" + sfx]); kps["angle"] = f32(c["angle" + sfx]); kps["octave"] = np.array(c["octave" + sfx], np.int32)
            d["kps" + sfx] = kps
            d["uright" + sfx] = f32(c["uright" + sfx])
            d["has_mp" + sfx] = np.array(c["has_mp" + sfx], np.uint8)
            d["desc" + sfx] = np.frombuffer(bytes.fromhex(c["desc" + sfx]), np.uint8).reshape(N, 32).copy()
            d["nodes" + sfx] = (np.array(c["node_id" + sfx], np.uint32), np.array(c["node_start" + sfx], np.int32), np.array(c["feat" + sfx], np.int32))
        out.append(d)
    return out


def load_grid_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    return [dict(x=f32(c["x"]), y=f32(c["y"]), bounds=tuple(float(v) for v in f32(c["bounds"])), cell_start=np.array(c["cell_start"], np.int32),
                 cell_idx=np.array(c["cell_idx"], np.int32)) for c in json.load(open(path))["cases"]]


def load_undistort_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    return [dict(n=c["n"], cam=f32(c["cam"]), x=f32(c["x"]), y=f32(c["y"]), x_un=f32(c["x_un"]), y_un=f32(c["y_un"]), copied=c["other_fields_copied"])
            for c in json.load(open(path))["cases"]]


def load_frustum_cases(path):
    import json
    f32 = lambda a: np.array(a, dtype=np.uint32).view(np.float32)
    out = []
    for c in json.load(open(path))["cases"]:
        M = c["m"]; cam = f32(c["cam"])
        out.append(dict(xw=f32(c["world_pos"]).reshape(M, 3).copy(), normal=f32(c["normal"]).reshape(M, 3).copy(), dmin=f32(c["min_dist"]), dmax=f32(c["max_dist"]),
                        Rcw=f32(c["Rcw"]).reshape(3, 3).copy(), tcw=f32(c["tcw"]), Ow=f32(c["Ow"]), cam=cam, in_view=np.array(c["in_view"], np.uint8),
                        proj_x=f32(c["proj_x"]), proj_y=f32(c["proj_y"]), proj_xr=f32(c["proj_xr"]), level=np.array(c["level"], np.int32), view_cos=f32(c["view_cos"])))
    return out


def stereo_case(seed=9801, w=640, h=480):
    """mirror of oracle/refprobe/probe.cpp tier J: the random depth image (float32) the reference's ComputeStereoFromRGBD was run on"""
    g = LCG(seed)
    f = np.float32
    def uf():
        return f(g.u32() >> 8) * f(1.0 / 16777216.0)
    depth = np.zeros(w * h, np.float32)
    for i in range(w * h):
        u = uf()
        depth[i] = f(0.0) if u < f(0.08) else (f(-1.0) if u < f(0.1) else f(0.4) + uf() * f(6.0))
    return depth.reshape(h, w)

