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
