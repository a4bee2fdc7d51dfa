"""Seeded synthetic RGB-D frames: mid-grey background + random filled convex
polygons (corners AND long straight edges) + small square marks + 2-octave value noise + Gaussian pixel noise + 3x3 box
blur.  Depth: plane + per-object offsets, TUM scale (x5000), 5 % zeros.  Pure numpy; used by
bench.py and the tests (there is no dataset on the GPU box)."""
import numpy as np


def _value_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-amp, amp, (gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def synth_frame(seed, w=640, h=480, with_depth=False):
    rng = np.random.Generator(np.random.PCG64(0xC0FFEE + int(seed)))
    img = np.full((h, w), 128.0, np.float32)
    depth = (1.0 + 2.0 * np.arange(h, dtype=np.float32)[:, None] / h + np.zeros((1, w), np.float32)) if with_depth else None
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    area = (w * h) / (640.0 * 480.0)
    npoly = int(rng.integers(140, 221) * area)
    for _ in range(npoly):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        rad = rng.uniform(0.015, 0.16) * 640
        nv = int(rng.integers(3, 7))
        ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
        px = cx + rad * np.cos(ang) * rng.uniform(0.6, 1.0, nv)
        py = cy + rad * np.sin(ang) * rng.uniform(0.6, 1.0, nv)
        x0, x1 = int(max(0, np.floor(px.min()))), int(min(w, np.ceil(px.max()) + 1))
        y0, y1 = int(max(0, np.floor(py.min()))), int(min(h, np.ceil(py.max()) + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        X = xx[y0:y1, x0:x1]; Y = yy[y0:y1, x0:x1]
        inside = np.ones(X.shape, bool)
        for i in range(nv):
            j = (i + 1) % nv
            inside &= ((px[j] - px[i]) * (Y - py[i]) - (py[j] - py[i]) * (X - px[i])) >= 0
        val = rng.uniform(0, 255)
        img[y0:y1, x0:x1][inside] = val
        if with_depth:
            depth[y0:y1, x0:x1][inside] = rng.uniform(0.5, 4.0)
    # small high-contrast marks (texture corners)
    nm = int(rng.integers(300, 501) * area)
    mx = rng.integers(0, w - 6, nm); my = rng.integers(0, h - 6, nm)
    ms = rng.integers(3, 7, nm); mv = rng.uniform(0, 255, nm)
    for i in range(nm):
        img[my[i]:my[i] + ms[i], mx[i]:mx[i] + ms[i]] = mv[i]
    img += _value_noise(rng, h, w, 32, 12.0) + _value_noise(rng, h, w, 16, 6.0)
    img += rng.normal(0, 2.0, (h, w)).astype(np.float32)
    p = np.pad(img, 1, mode="edge")
    img = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9.0
    gray = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if not with_depth:
        return gray
    d16 = np.clip(depth * 5000.0, 0, 65535).astype(np.uint16)
    d16[rng.uniform(0, 1, (h, w)) < 0.05] = 0
    return gray, d16


def synth_batch(seed0, n, w=640, h=480):
    return np.stack([synth_frame(seed0 + i, w, h) for i in range(n)])
