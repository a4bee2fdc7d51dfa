"""REAL photographs through the path (VERDICT r05, weak 2 / missing 3: the TUM-named configurations had only ever run on synthetic stand-ins, and no TUM image
exists in this image or on the GPU box).  tests/golden/real holds seven CC0 / public-domain photographs (a camera man, an astronaut portrait, a coffee cup, a cat,
bricks, grass, gravel -- tests/golden/make_real_photos.py, MANIFEST.json with provenance, licences and Pillow's decode of each file).
CPU: the in-tree PNG reader against Pillow's pixels, the photo family of synth.py, the oracle on real input.
GPU: ORB and LSD+LBD of every photograph at its native size, of VGA windows of them (one frame, 8 in flight, a large batch) and the RGB ingest in front, byte for
byte against the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

import orc
from conftest import gpu_available

REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real")
MAN = json.load(open(os.path.join(REAL, "MANIFEST.json")))


def _need_gpu():
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")


def _eq_orb(kps, desc, ref):
    assert len(kps) == len(ref["kps"])
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kps[f].view(np.uint32), ref["kps"][f].view(np.uint32)), f
    assert np.array_equal(kps["octave"], ref["kps"]["octave"]) and np.array_equal(desc, ref["desc"])


def _eq_lines(got, ref, what):
    kl, desc, eq = got
    assert kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"]) and np.array_equal(eq.view(np.uint64), ref["eq"].view(np.uint64)), what


def test_png_reader_equals_pillow_on_real_files():
    """every byte of every photograph as libpng (Pillow) decodes it: the pinned hash always, Pillow itself where it is installed"""
    from rgbd_pl_slam_amd.png import read_png
    assert len(MAN) == 7
    for name, m in MAN.items():
        path = os.path.join(REAL, name)
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == m["sha256"], name
        im = read_png(path)
        assert list(im.shape) == m["shape"] and str(im.dtype) == m["dtype"], name
        assert hashlib.sha256(np.ascontiguousarray(im).tobytes()).hexdigest() == m["pixels_sha256"], name
    Image = pytest.importorskip("PIL.Image")
    for name in MAN:
        assert np.array_equal(read_png(os.path.join(REAL, name)), np.asarray(Image.open(os.path.join(REAL, name)))), name


def test_photo_family_frames():
    from rgbd_pl_slam_amd.synth import photo_frame, photos, synth_batch_parallel
    ph = photos()
    assert len(ph) == 7 and all(p.ndim == 2 and p.dtype == np.uint8 for p in ph)
    # the colour photographs went through the same 8-bit RGB -> gray arithmetic as the oracle's ingest (orc_rgb_to_gray, R first)
    from rgbd_pl_slam_amd.png import read_png
    names = sorted(MAN)
    k = names.index("coffee.png")
    assert np.array_equal(ph[k], orc.rgb_to_gray(np.ascontiguousarray(read_png(os.path.join(REAL, "coffee.png"))[:, :, :3]), bgr=False))
    a, b = photo_frame(5), photo_frame(5)
    assert a.shape == (480, 640) and a.dtype == np.uint8 and np.array_equal(a, b) and not np.array_equal(a, photo_frame(12))   # (5 and 12: the same photograph)
    assert photo_frame(3, 1280, 960).shape == (960, 1280) and photo_frame(2, 320, 240).shape == (240, 320)
    batch = synth_batch_parallel(100, 40, 640, 480, family="photo")
    assert batch.shape == (40, 480, 640) and np.array_equal(batch[7], photo_frame(107))
    # a window that fits the photograph is a plain crop of it (no resampling): it must be found in the photograph or its mirror image
    f = photo_frame(0, 200, 160)
    g = ph[0]
    hit = False
    for src in (g, g[:, ::-1]):
        for y in range(src.shape[0] - 160 + 1):
            rows = np.flatnonzero((src[y, :src.shape[1] - 199] == f[0, 0]))
            for x in rows:
                if np.array_equal(src[y:y + 160, x:x + 200], f):
                    hit = True
                    break
            if hit:
                break
        if hit:
            break
    assert hit


def test_oracle_on_real_photographs():
    """the oracle finds the full feature budget on real texture: 1000 key points on every photograph, the best 100 lines on every photograph with man-made edges"""
    from rgbd_pl_slam_amd.synth import photos
    names = sorted(MAN)
    for name, g in zip(names, photos()):
        o = orc.orb_extract(g)
        l = orc.line_extract(g, 100)
        assert 1000 <= len(o["kps"]) <= 1100, name           # (octree distribution: nfeatures + the per-level surplus)
        assert len(l["kl"]) == 100 or name in ("grass.png", "gravel.png", "chelsea.png"), (name, len(l["kl"]))
        assert len(l["kl"]) >= 20, (name, len(l["kl"]))


@pytest.mark.gpu
def test_real_photographs_native_size_exact():
    _need_gpu()
    from rgbd_pl_slam_amd import ORBextractor, LineSegment
    from rgbd_pl_slam_amd.synth import photos
    names = sorted(MAN)
    ext = ORBextractor(nfeatures=1000, max_width=640, max_height=512, max_batch=1)
    ls = LineSegment(nlines=100, max_width=640, max_height=512, max_batch=1)
    for name, g in zip(names, photos()):
        kps, desc = ext(g)
        _eq_orb(kps, desc, orc.orb_extract(g))
        _eq_lines(ls.ExtractLineSegment(g), orc.line_extract(g, 100), name)
    ext.close(); ls.close()


@pytest.mark.gpu
def test_real_photographs_rgb_ingest_then_extract():
    """GrabImageRGBD on the colour photographs: RGB -> gray on the GPU (both channel orders), then both extractors on the GPU's gray image"""
    _need_gpu()
    import torch
    from rgbd_pl_slam_amd import ORBextractor, LineSegment, frame
    from rgbd_pl_slam_amd.png import read_png
    for name in ("astronaut.png", "coffee.png", "chelsea.png"):
        rgb = np.ascontiguousarray(read_png(os.path.join(REAL, name))[:, :, :3])
        h, w = rgb.shape[:2]
        for bgr in (False, True):
            d = torch.from_numpy(rgb[None]).cuda()
            g = torch.zeros((1, h, w), dtype=torch.uint8, device="cuda")
            frame.rgb_to_gray(d, g, bgr_order=bgr)
            torch.cuda.synchronize()
            gh = g[0].cpu().numpy()
            assert np.array_equal(gh, orc.rgb_to_gray(rgb, bgr)), (name, bgr)
        ext = ORBextractor(nfeatures=2000, max_width=w, max_height=h, max_batch=1)
        ls = LineSegment(nlines=200, max_width=w, max_height=h, max_batch=1)
        kps, desc = ext(gh)
        _eq_orb(kps, desc, orc.orb_extract(gh, nfeatures=2000))
        _eq_lines(ls.ExtractLineSegment(gh), orc.line_extract(gh, 200), name)
        ext.close(); ls.close()


@pytest.mark.gpu
def test_real_photo_windows_one_few_and_many_in_flight():
    """VGA windows of the photographs (synth.photo_frame): one frame at a time, BASELINE configs[2]'s 8 frames in flight, and 96 / 700 in flight (the mid-range and
    the one-wave-per-frame schedules), every frame against the oracle"""
    _need_gpu()
    from concurrent.futures import ThreadPoolExecutor
    from rgbd_pl_slam_amd import ORBextractor, LineSegment
    from rgbd_pl_slam_amd.synth import photo_frame
    pool = ThreadPoolExecutor(16)
    n = 28
    imgs = list(pool.map(lambda s: photo_frame(500 + s), range(n)))
    refs_o = list(pool.map(lambda im: orc.orb_extract(im), imgs))
    refs_l = list(pool.map(lambda im: orc.line_extract(im, 100), imgs))
    ext = ORBextractor(nfeatures=1000, max_width=640, max_height=480, max_batch=700)
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=700)
    for k in range(7):
        kps, desc = ext(imgs[k])
        _eq_orb(kps, desc, refs_o[k])
        _eq_lines(ls.ExtractLineSegment(imgs[k]), refs_l[k], k)
    for B in (8, 96, 700):
        idx = [(3 * i) % n for i in range(B)]
        stack = np.stack([imgs[i] for i in idx])
        ro = ext.extract_batch(stack); rl = ls.extract_batch(stack)
        for j, i in enumerate(idx):
            _eq_orb(ro[j][0], ro[j][1], refs_o[i])
            _eq_lines(rl[j], refs_l[i], (B, j))
    ext.close(); ls.close()


@pytest.mark.gpu
def test_real_photo_windows_1280x960_config4():
    """BASELINE configs[3]'s frame size and budgets (1280x960, 4000 + 400) on windows of the photographs extended by reflection, 8 in flight"""
    _need_gpu()
    from concurrent.futures import ThreadPoolExecutor
    from rgbd_pl_slam_amd import ORBextractor, LineSegment
    from rgbd_pl_slam_amd.synth import photo_frame
    pool = ThreadPoolExecutor(8)
    imgs = list(pool.map(lambda s: photo_frame(900 + s, 1280, 960), range(8)))
    refs_o = list(pool.map(lambda im: orc.orb_extract(im, nfeatures=4000), imgs))
    refs_l = list(pool.map(lambda im: orc.line_extract(im, 400), imgs))
    ext = ORBextractor(nfeatures=4000, max_width=1280, max_height=960, max_batch=8)
    ls = LineSegment(nlines=400, max_width=1280, max_height=960, max_batch=8)
    ro = ext.extract_batch(np.stack(imgs)); rl = ls.extract_batch(np.stack(imgs))
    for k in range(8):
        _eq_orb(ro[k][0], ro[k][1], refs_o[k])
        _eq_lines(rl[k], refs_l[k], k)
    ext.close(); ls.close()
