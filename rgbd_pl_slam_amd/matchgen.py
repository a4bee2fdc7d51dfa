"""Synthetic matching workloads shared by bench.py, the soak tools and the tests (tests/matchgen.py re-exports this module)."""
import numpy as np


def flip_bits(desc, rng, maxflips):
    out = desc.copy()
    for i in range(len(out)):
        k = int(rng.integers(0, maxflips + 1))
        if k:
            pos = rng.choice(256, k, replace=False)
            for p in pos:
                out[i, p >> 3] ^= np.uint8(1 << (p & 7))
    return out


def make_local_map(kps, desc, M, seed, w=640, h=480, nlevels=8, with_uright=False, uright=None):
    """kps: structured keypoints of the current frame (x, y, octave); returns dict of numpy arrays describing M map points.
    uright (the frame's mvuRight): map points derived from a key point with stereo data get a consistent mTrackProjXR"""
    rng = np.random.default_rng(seed)
    N = len(kps)
    src = rng.integers(0, N, M)
    real = rng.uniform(0, 1, M) < 0.6
    px = np.where(real, kps["x"][src] + rng.normal(0, 2.0, M), rng.uniform(-40, w + 40, M)).astype(np.float32)
    py = np.where(real, kps["y"][src] + rng.normal(0, 2.0, M), rng.uniform(-40, h + 40, M)).astype(np.float32)
    lvl = np.where(real, np.clip(kps["octave"][src] + rng.integers(0, 2, M), 0, nlevels - 1), rng.integers(0, nlevels, M)).astype(np.int32)
    d = desc[src].copy()
    d[real] = flip_bits(d[real], rng, 40)
    d[~real] = rng.integers(0, 256, (int((~real).sum()), 32), dtype=np.uint8)
    view_cos = rng.uniform(0.99, 1.0, M).astype(np.float32)
    in_view = (rng.uniform(0, 1, M) < 0.9).astype(np.uint8)
    proj_xr = (px - rng.uniform(2, 30, M)).astype(np.float32)
    obs_positive = (rng.uniform(0, 1, M) < 0.97).astype(np.uint8)
    if uright is not None:
        ur = np.asarray(uright, np.float32)[src]
        proj_xr = np.where(real & (ur > 0), ur + rng.normal(0, 1.0, M), proj_xr).astype(np.float32)
    return dict(proj_x=px, proj_y=py, proj_xr=proj_xr, level=lvl, view_cos=view_cos, in_view=in_view, desc=d, obs_positive=obs_positive)


def make_map_lines(kl, ldesc, M, seed, nlevels=8):
    rng = np.random.default_rng(seed)
    N = len(kl)
    src = rng.integers(0, N, M)
    real = rng.uniform(0, 1, M) < 0.6
    jit = rng.normal(0, 1.5, (M, 4)).astype(np.float32)
    x1 = np.where(real, kl["startPointX"][src] + jit[:, 0], rng.uniform(0, 640, M)).astype(np.float32)
    y1 = np.where(real, kl["startPointY"][src] + jit[:, 1], rng.uniform(0, 480, M)).astype(np.float32)
    x2 = np.where(real, kl["endPointX"][src] + jit[:, 2], rng.uniform(0, 640, M)).astype(np.float32)
    y2 = np.where(real, kl["endPointY"][src] + jit[:, 3], rng.uniform(0, 480, M)).astype(np.float32)
    d = ldesc[src].copy()
    d[real] = flip_bits(d[real], rng, 30)
    d[~real] = rng.integers(0, 256, (int((~real).sum()), 32), dtype=np.uint8)
    return dict(x1=x1, y1=y1, x2=x2, y2=y2, level=rng.integers(0, 2, M).astype(np.int32), view_cos=rng.uniform(0.99, 1.0, M).astype(np.float32),
                in_view=(rng.uniform(0, 1, M) < 0.9).astype(np.uint8), desc=d)


def make_last_frame(kps, desc, seed, fx=525.0, fy=525.0, cx=319.5, cy=239.5, bf=40.0):
    """a "last frame" whose map points reproject close to the key points `kps` under a small forward motion: the input of
    ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono).  Returns (last dict of numpy arrays, pose dict)."""
    rng = np.random.default_rng(seed)
    n = len(kps)
    z = rng.uniform(0.8, 4.0, n).astype(np.float32)
    # last camera at the origin; current camera moved 2 cm forward and 1 cm sideways
    xw = np.stack([(kps["x"] - cx) * z / fx, (kps["y"] - cy) * z / fy, z], 1).astype(np.float32)
    pose = dict(Rcw=np.eye(3, dtype=np.float32), tcw=np.array([0.01, -0.005, -0.02], np.float32), Rlw=np.eye(3, dtype=np.float32),
                tlw=np.zeros(3, np.float32), fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, b=bf / fx)
    last = dict(keys=kps.copy(), has_mappoint=(rng.uniform(0, 1, n) < 0.8).astype(np.uint8), outlier=(rng.uniform(0, 1, n) < 0.05).astype(np.uint8),
                world_pos=xw, mp_desc=flip_bits(desc, rng, 20), obs_positive=np.ones(n, np.uint8))
    return last, pose
