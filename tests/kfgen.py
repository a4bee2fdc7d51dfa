"""Random keyframe / map-point scenes for the LocalMapping / LoopClosing matcher overloads (test data only).

The golden fixtures (tests/golden/ref_glue_{fuse,fuse_sim3,search_sim3,sim3,triangulation}.json) pin the oracle to the
reference binary on three cases each; these scenes widen the GPU-vs-oracle comparison to other sizes, densities and poses."""
import numpy as np

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
F = np.float32


def _rot(ay, ax):
    cy, sy, cx, sx = np.cos(ay), np.sin(ay), np.cos(ax), np.sin(ax)
    return np.array([[cy, sy * sx, sy * cx], [0, cx, -sx], [-sy, cy * sx, cy * cx]], dtype=np.float32)


def scales(n=8, f=1.2):
    s = np.ones(n, np.float32)
    for i in range(1, n):
        s[i] = np.float32(np.float64(s[i - 1]) * f)
    return s


def keyframe_scene(seed, nk, m, sim3_scale=None, width=640, height=480):
    """keyframe with nk key points and m map points, most of them re-observations of a key point.
    Returns dict(kps, uright, desc, scale, pose, pts, Scw)."""
    r = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = F(517.3), F(516.5), F(318.6), F(255.3), F(40.0)
    sc = scales()
    kps = np.zeros(nk, KP_DTYPE)
    kps["x"] = r.uniform(0, width, nk).astype(F); kps["y"] = r.uniform(0, height, nk).astype(F)
    kps["octave"] = r.integers(0, 8, nk); kps["angle"] = r.uniform(0, 360, nk).astype(F); kps["size"] = 31; kps["response"] = 1; kps["class_id"] = -1
    kz = r.uniform(0.6, 7.6, nk).astype(F)
    ur = np.where(r.random(nk) < 0.7, kps["x"] - bf / kz + r.uniform(-0.75, 0.75, nk).astype(F), F(-1)).astype(F)
    desc = r.integers(0, 256, (nk, 32), dtype=np.uint8)
    R = _rot(r.uniform(-0.08, 0.08), r.uniform(-0.05, 0.05)); t = r.uniform(-0.3, 0.3, 3).astype(F)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(F)
    src = r.integers(0, nk, m); tied = r.random(m) < 0.85
    u0 = np.where(tied, kps["x"][src] + r.uniform(-2.5, 2.5, m) * sc[kps["octave"][src]], r.uniform(-80, width + 80, m))
    v0 = np.where(tied, kps["y"][src] + r.uniform(-2.5, 2.5, m) * sc[kps["octave"][src]], r.uniform(-60, height + 60, m))
    z = np.where(tied, kz[src] * (1 + r.uniform(-0.01, 0.01, m)), r.uniform(0.5, 8.5, m))
    z = np.where(r.random(m) < 0.03, -z, z)
    Xc = np.stack([(u0 - cx) / fx * z, (v0 - cy) / fy * z, z], 1)
    xw = ((Xc - t) @ R.astype(np.float64)).astype(F)          # R^T (Xc - t)
    PO = xw - Ow; dist = np.linalg.norm(PO, axis=1)
    nrm = PO / dist[:, None] + r.uniform(-0.3, 0.3, (m, 3)); flip = r.random(m) < 0.08; nrm[flip] = -nrm[flip]
    nrm = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(F)
    plev = np.where(tied, kps["octave"][src], r.integers(0, 8, m))
    dmax = (dist * 1.2 ** (plev + (r.random(m) < 0.5) - 0.5 + r.uniform(-0.45, 0.45, m))).astype(F)
    dmin = (dmax / F(1.2 ** 7)).astype(F)
    far = r.random(m) < 0.05; dmax[far] = (dist[far] * 0.6).astype(F); dmin[far] = dmax[far] / 4
    mdesc = np.where(tied[:, None], desc[src], r.integers(0, 256, (m, 32), dtype=np.uint8)).astype(np.uint8)
    for i in range(m):
        bits = r.integers(0, 256, r.integers(0, 80))
        for b in bits:
            mdesc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    valid = (r.random(m) > 0.08).astype(np.uint8)
    is2 = (1.0 / (sc.astype(np.float64) ** 2)).astype(F)
    pose = dict(Rcw=R, tcw=t, Ow=Ow, fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), bf=float(bf), log_scale_factor=float(F(np.log(F(1.2)))), inv_sigma2=is2)
    out = dict(kps=kps, uright=ur, desc=desc, scale=sc, pose=pose, bounds=(0.0, 0.0, float(width), float(height)),
               pts=dict(xw=xw, normal=nrm, min_dist=dmin, max_dist=dmax, desc=mdesc, valid=valid))
    if sim3_scale is not None:
        S = np.eye(4, dtype=F); S[:3, :3] = F(sim3_scale) * R; S[:3, 3] = F(sim3_scale) * t
        out["Scw"] = S
        out["intr"] = {k: pose[k] for k in ("fx", "fy", "cx", "cy", "bf", "log_scale_factor")}
        out["init"] = np.where(r.random(nk) < 0.25, -2, -1).astype(np.int32)
    return out


def two_keyframe_scene(seed, n, nodes=100, s12=1.0, tz=0.2):
    """two keyframes observing the same n world points (key point i of KF1 <-> key point perm[i] of KF2), with map points, uRight,
    a vocabulary node per feature and the true fundamental matrix.  Returns a dict usable by orc.search_by_sim3 /
    orc.search_for_triangulation and the HIP wrappers."""
    r = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = 517.3, 516.5, 318.6, 255.3, 40.0
    sc = scales(); sig2 = (sc.astype(np.float64) ** 2).astype(F)
    R1 = _rot(0.02, -0.01).astype(np.float64); t1 = np.zeros(3)
    R2 = _rot(-0.03, 0.02).astype(np.float64); t2 = np.array([-0.12, 0.03, tz])
    perm = r.permutation(n)
    u1 = r.uniform(10, 630, n); v1 = r.uniform(10, 470, n); z1 = r.uniform(0.8, 6.8, n)
    Xc1 = np.stack([(u1 - cx) / fx * z1, (v1 - cy) / fy * z1, z1], 1)
    Xw = (Xc1 - t1) @ R1
    Xc2 = Xw @ R2.T + t2
    u2 = fx * Xc2[:, 0] / Xc2[:, 2] + cx; v2 = fy * Xc2[:, 1] / Xc2[:, 2] + cy
    o1 = r.integers(0, 8, n); o2 = np.clip(o1 + r.integers(-1, 2, n), 0, 7)
    noise = np.where(r.random(n) < 0.8, 1.5, 12.0) * sc[o2]
    k1 = np.zeros(n, KP_DTYPE); k2 = np.zeros(n, KP_DTYPE)
    k1["x"] = u1; k1["y"] = v1; k1["octave"] = o1; k1["angle"] = r.uniform(0, 360, n)
    a2 = k1["angle"] - np.where(r.random(n) < 0.8, r.uniform(20, 28, n), r.uniform(0, 360, n)); a2 = np.where(a2 < 0, a2 + 360, a2)
    k2["x"][perm] = u2 + r.uniform(-0.5, 0.5, n) * noise; k2["y"][perm] = v2 + r.uniform(-0.5, 0.5, n) * noise
    k2["octave"][perm] = o2; k2["angle"][perm] = a2
    d1 = r.integers(0, 256, (n, 32), dtype=np.uint8); d2 = np.zeros_like(d1)
    for i in range(n):
        d = d1[i].copy()
        for b in r.integers(0, 256, r.integers(0, 64)):
            d[b >> 3] ^= np.uint8(1 << (b & 7))
        d2[perm[i]] = d
    ur1 = np.where(r.random(n) < 0.5, u1 - bf / z1, -1).astype(F); ur2 = np.full(n, -1, F)
    st = r.random(n) < 0.5; ur2[perm[st]] = (k2["x"][perm[st]] - bf / Xc2[st, 2]).astype(F)
    has1 = (r.random(n) < 0.35).astype(np.uint8); has2 = (r.random(n) < 0.35).astype(np.uint8)
    nd1 = (700 + 2 * r.integers(0, nodes, n)).astype(np.uint32); nd2 = np.zeros(n, np.uint32)
    nd2[perm] = np.where(r.random(n) < 0.9, nd1, 700 + 2 * r.integers(0, nodes, n) + (r.random(n) < 0.3))
    def flat(nd):
        ids = np.unique(nd); starts = [0]; feats = []
        for k in ids:
            f = np.nonzero(nd == k)[0]; feats += list(f); starts.append(len(feats))
        return ids.astype(np.uint32), np.array(starts, np.int32), np.array(feats, np.int32)
    R12 = R1 @ R2.T; t12 = t1 - R12 @ t2
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]]); Ki = np.linalg.inv(K)
    F12 = (Ki.T @ tx @ R12 @ Ki).astype(F)
    lsf = float(F(np.log(F(1.2))))
    def pose(R, t):
        return dict(Rcw=R.astype(F), tcw=t.astype(F), Ow=(-(R.T @ t)).astype(F), fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, log_scale_factor=lsf, inv_sigma2=np.ones(8, F))
    def pts(idx_of_world, dist_other, oct_other, desc, has):
        m = len(desc)
        xw = np.zeros((m, 3), F); xw[idx_of_world] = (Xw + r.uniform(-0.005, 0.005, Xw.shape)).astype(F)
        dmax = np.zeros(m, F); dmax[idx_of_world] = (dist_other * 1.2 ** (oct_other + (r.random(n) < 0.5) - 0.5 + r.uniform(-0.45, 0.45, n))).astype(F)
        md = desc.copy()
        return dict(xw=xw, normal=np.zeros((m, 3), F), min_dist=(dmax / F(1.2 ** 7)).astype(F), max_dist=dmax, desc=md,
                    valid=(has & (r.random(m) > 0.05)).astype(np.uint8))
    ar = np.arange(n)
    c = dict(n=n, scale=sc, sigma2=sig2, F12=F12, kps1=k1, kps2=k2, desc1=d1, desc2=d2, uright1=ur1, uright2=ur2, has_mp1=has1, has_mp2=has2,
             nodes1=flat(nd1), nodes2=flat(nd2), Ow1=(-(R1.T @ t1)).astype(F), R2w=R2.astype(F), t2w=t2.astype(F), fx=fx, fy=fy, cx=cx, cy=cy,
             pose1=pose(R1, t1), pose2=pose(R2, t2), s12=float(s12), R12=R12.astype(F), t12=(t1 - s12 * (R12 @ t2)).astype(F),
             pts1=pts(ar, np.linalg.norm(Xc2, axis=1), o2, d1, (r.random(n) < 0.85).astype(np.uint8)),
             pts2=pts(perm, np.linalg.norm(Xc1, axis=1), o1, d2, (r.random(n) < 0.85).astype(np.uint8)))
    return c
