"""GPU parity: HIP LSD + LBD line extractor (through the C ABI) vs the CPU oracle.
Segments / KeyLine floats: every arithmetic step is the same IEEE operation sequence as the oracle except the
double-precision libm calls (cos/sin/atan2/log/exp/pow), whose last-bit differences are far below float rounding;
the test therefore demands bit-equal float outputs and allows 1e-4 (north-star tolerance) only as a reported fallback.
LBD bytes: exact."""
import numpy as np
import pytest

import orc
from conftest import gpu_available

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _need_gpu():
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")


# Few frames in flight take the banded speculative schedule.  Its second half exists in three forms, all of which must give the serial result:
#   "rounds"     validation rounds (every band validates itself, all at once) until the fixpoint -- the default up to 16 frames;
#   "one_round"  a single round, which rarely reaches the fixpoint: the serial commit wave then finishes the frame from the half-validated logs;
#   "commit"     no rounds: the one-launch schedule with the serial commit wave (what more than 16 frames in flight take).
SCHEDULES = pytest.mark.parametrize("schedule", ["rounds", "one_round", "commit", "two_launch", "fill"])


def _set_schedule(monkeypatch, schedule):
    if schedule == "one_round":
        monkeypatch.setenv("PLF_LSD_SPEC_ROUNDS", "1")
    elif schedule == "commit":
        monkeypatch.setenv("PLF_LSD_SPEC_Z", "0")
    elif schedule == "two_launch":   # band waves and serial commit wave as two launches (k_lsd_spec_grow + k_lsd_spec_commit): what batches beyond the fused
        monkeypatch.setenv("PLF_LSD_SPEC_Z", "0")          # launch's residency limit take when the validation rounds are off (round 6: they are on up to 256 frames in flight)
        monkeypatch.setenv("PLF_LSD_SPEC_NOFUSE", "1")
    elif schedule == "fill":       # validation rounds on the no-growth guess of the state above a band (PLF_LSD_SPEC_FILL, off by default)
        monkeypatch.setenv("PLF_LSD_SPEC_FILL", "8")


def _check(img, nlines, ext=None, lbd_sobel_input=0):
    from rgbd_pl_slam_amd import LineSegment
    h, w = img.shape
    own = ext is None
    if own:
        ext = LineSegment(nlines=nlines, max_width=w, max_height=h, lbd_sobel_input=lbd_sobel_input)
    kl, desc, eq = ext.ExtractLineSegment(img)
    segs = ext.segments(0)
    ref_seg = orc.lsd_detect(img)["lines"]
    ref = orc.line_extract(img, nlines, lbd_sobel_input=lbd_sobel_input)
    assert len(segs) == len(ref_seg), "segment count %d vs %d" % (len(segs), len(ref_seg))
    assert np.allclose(segs, ref_seg, rtol=0, atol=TOL)
    nbits = int((segs.view(np.uint32) != ref_seg.view(np.uint32)).sum())
    assert len(kl) == len(ref["kl"])
    for name in kl.dtype.names:
        a, b = kl[name], ref["kl"][name]
        if a.dtype.kind == "f":
            assert np.allclose(a, b, rtol=0, atol=TOL * max(1.0, float(np.abs(b).max()) if len(b) else 1.0)), name
        else:
            assert np.array_equal(a, b), name
    assert np.allclose(eq, ref["eq"], rtol=0, atol=1e-4)
    same_fields = all(np.array_equal(kl[n].view(np.uint32), ref["kl"][n].view(np.uint32)) for n in kl.dtype.names)
    bad_rows = int((desc != ref["desc"]).any(1).sum())
    assert bad_rows == 0, "LBD descriptors differ in %d of %d rows (keylines bit-equal: %s)" % (bad_rows, len(desc), same_fields)
    assert nbits == 0 and same_fields, "float outputs within 1e-4 but not bit-equal (segments differing words: %d)" % nbits
    if own:
        ext.close()


def test_lines_vga_synthetic():
    _need_gpu()
    from rgbd_pl_slam_amd.synth import synth_frame
    for seed in (0, 1):
        _check(synth_frame(seed), 100)


@pytest.mark.parametrize("w,h", [(640, 480), (322, 243), (641, 479), (1280, 960)])
def test_lbd_sobel_input_switch_both_settings(w, h):
    """plf_line_params.lbd_sobel_input: BinaryDescriptor's Sobel on octave 0 of its Gaussian pyramid (GaussianBlur 5x5 sigma 1, the default,
    believed opencv_contrib 3.3 behaviour) or on the raw image -- both bit-exact against the oracle, and they really differ.  Widths with
    w % 4 != 0 exercise the scalar tail rounding of the column filter, odd heights the REFLECT_101 border of blur and Sobel."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    img = synth_frame(11, w, h)
    _check(img, 100, lbd_sobel_input=orc.LBD_BLURRED)
    _check(img, 100, lbd_sobel_input=orc.LBD_RAW)
    a = orc.line_extract(img, 100, lbd_sobel_input=orc.LBD_BLURRED); b = orc.line_extract(img, 100, lbd_sobel_input=orc.LBD_RAW)
    assert np.array_equal(a["kl"], b["kl"])                 # detection does not depend on the switch
    assert (a["desc"] != b["desc"]).any(1).mean() > 0.5     # the descriptors do


def test_lines_fewer_than_requested_and_200():
    _need_gpu()
    from rgbd_pl_slam_amd.synth import synth_frame
    img = synth_frame(2)
    _check(img, 200)
    _check(img, 1000)   # fewer segments than nlines: detection order, no sort


def test_lines_odd_size_and_large():
    _need_gpu()
    from rgbd_pl_slam_amd.synth import synth_frame
    _check(synth_frame(4, 752, 480), 100)
    _check(synth_frame(5, 1280, 960), 400)


def test_lines_flat_and_noise():
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    ext = LineSegment(nlines=100, max_width=640, max_height=480)
    kl, desc, eq = ext.ExtractLineSegment(np.full((480, 640), 99, np.uint8))
    assert len(kl) == 0
    rng = np.random.default_rng(3)
    _check(rng.integers(0, 256, (480, 640), dtype=np.uint8), 100, ext=ext)
    ext.close()


def test_lines_batch_equals_single():
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_batch
    imgs = synth_batch(20, 3)
    ext = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=3)
    res = ext.extract_batch(imgs)
    for f in range(3):
        ref = orc.line_extract(imgs[f], 100)
        assert np.array_equal(res[f][1], ref["desc"])
        assert np.array_equal(res[f][0]["startPointX"].view(np.uint32), ref["kl"]["startPointX"].view(np.uint32))
    ext.close()


def test_lines_published_seed_order():
    """seed_order = 1 (bins descending, raster inside a bin: every OpenCV release except 3.0-3.3) on the GPU == the oracle
    run with the same switch, bit for bit, single frame and inside a batch; and it differs from the default order."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame, synth_batch
    for img in (synth_frame(21), synth_frame(22, 320, 240), synth_frame(23, 752, 480)):
        h, w = img.shape
        ext = LineSegment(nlines=100, max_width=w, max_height=h, seed_order=1)
        kl, desc, eq = ext.ExtractLineSegment(img)
        segs = ext.segments(0)
        ref_seg = orc.lsd_detect(img, seed_order=1)["lines"]
        ref = orc.line_extract(img, 100, seed_order=1)
        assert len(segs) == len(ref_seg) and np.array_equal(segs.view(np.uint32), ref_seg.view(np.uint32))
        assert len(kl) == len(ref["kl"]) and np.array_equal(desc, ref["desc"])
        for name in kl.dtype.names:
            assert np.array_equal(kl[name].view(np.uint32), ref["kl"][name].view(np.uint32)), name
        ext.close()
    imgs = synth_batch(30, 6)
    ext = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=6, seed_order=1)
    res = ext.extract_batch(imgs)
    differs = False
    for f in range(6):
        ref = orc.line_extract(imgs[f], 100, seed_order=1)
        assert np.array_equal(res[f][1], ref["desc"])
        differs = differs or not np.array_equal(orc.lsd_detect(imgs[f], seed_order=1)["lines"], orc.lsd_detect(imgs[f])["lines"])
    assert differs   # the two seed orders are genuinely different detectors
    ext.close()


@pytest.mark.parametrize("seed_order", [0, 1])
def test_lines_large_batch_kernel_on_a_small_batch(monkeypatch, seed_order):
    """k_lsd_regions2 (the throughput launch: 8 frames per workgroup, seed chunk parked in LDS) normally needs > 640 frames in flight (seed_order 0)
    or > 8 (seed_order 1, which has no speculative schedule); with the speculative schedule switched off it runs for 21 frames -- a count that leaves
    the last workgroup partly empty -- of mixed textures, each compared with the oracle run with the same seed order"""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import texture_frame
    monkeypatch.setenv("PLF_LSD_SPEC_MAX", "0")
    imgs = []
    s = 400
    while len(imgs) < 21:
        im, _ = texture_frame(s, size=(640, 480)); s += 1
        imgs.append(im)
    ext = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=21, seed_order=seed_order)
    res = ext.extract_batch(np.stack(imgs))
    for f in range(21):
        ref = orc.line_extract(imgs[f], 100, seed_order=seed_order)
        assert res[f][0].tobytes() == ref["kl"].tobytes() and np.array_equal(res[f][1], ref["desc"]), "frame %d" % f
    ext.close()


@pytest.mark.parametrize("w,h", [(12, 40), (40, 12), (15, 15), (16, 16), (17, 33), (19, 64), (64, 19)])
def test_lines_tiny_images(w, h):
    """images narrower than the 16-byte border window of k_lsd_pre's row pass (per-tap fallback) and images where EVERY 4-column group is a border group"""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    img = (((xx // 5 + yy // 7) & 1) * 180 + 30 + rng.integers(0, 8, (h, w))).astype(np.uint8)
    ls = LineSegment(nlines=20, max_width=max(w, 16), max_height=max(h, 16))
    kl, desc, eq = ls.ExtractLineSegment(img)
    ref = orc.line_extract(img, 20)
    assert len(kl) == len(ref["kl"]) > 0 and kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"])
    ls.close()


@pytest.mark.parametrize("h", [29, 30, 31, 59, 60, 61, 64, 65, 96, 97, 121])
def test_lines_heights_around_the_tile_rows_of_the_pre_pass(h):
    """k_lsd_pre works on 64 x 24 tiles of the 0.8x scaled image and k_blur5_sobel3 on 64 x 32 tiles of the input, with a copy of the body for interior tiles
    (no row reflection): heights whose scaled size ends just before / on / just after a tile row (24, 48 scaled rows = 30, 60 input rows; 32, 64, 96 input rows),
    both LBD inputs, both seed orders (the published order reads modgrad, which the pre-pass only writes where the level-line angle is defined)."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    rng = np.random.default_rng(100 + h)
    w = 200
    yy, xx = np.mgrid[0:h, 0:w]
    img = (((xx // 9 + yy // 6) & 1) * 150 + ((xx + 2 * yy) % 40) + 20 + rng.integers(0, 10, (h, w))).astype(np.uint8)
    for sobel_in in (0, 1):
        for seed_order in (0, 1):
            ext = LineSegment(nlines=40, max_width=w, max_height=h, lbd_sobel_input=sobel_in, seed_order=seed_order)
            kl, desc, eq = ext.ExtractLineSegment(img)
            segs = ext.segments(0)
            ref_seg = orc.lsd_detect(img, seed_order=seed_order)["lines"]
            ref = orc.line_extract(img, 40, lbd_sobel_input=sobel_in, seed_order=seed_order)
            assert len(segs) == len(ref_seg) and np.array_equal(segs.view(np.uint32), ref_seg.view(np.uint32)), (h, sobel_in, seed_order)
            assert len(kl) == len(ref["kl"]) and kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"]), (h, sobel_in, seed_order)
            ext.close()


def test_line_and_matcher_errors():
    """argument errors come back as PLF_E_BADARG (never a crash, never a silent wrong answer); an empty image is the
    reference's silent return (PLF_E_EMPTY, outputs untouched)"""
    _need_gpu()
    import ctypes as C
    import torch
    from rgbd_pl_slam_amd import LineSegment, Matcher, PlfError
    import rgbd_pl_slam_amd._lib as L
    with pytest.raises(PlfError):
        LineSegment(nlines=100, max_width=640, max_height=480, seed_order=2)
    ls = LineSegment(nlines=100, max_width=320, max_height=240)
    with pytest.raises(PlfError):
        ls.ExtractLineSegment(np.zeros((480, 640), np.uint8))        # larger than the handle was created for
    n = C.c_int32(-5)
    st = L.lib().plf_line_extract(ls._h, None, 0, 0, C.c_ssize_t(0), None, None, None, 100, C.byref(n))
    assert st == L.PLF_E_EMPTY and n.value == -5
    ls.close()
    m = Matcher(max_keypoints=256, max_mappoints=64, max_batch=2)
    z = torch.zeros(16, dtype=torch.int32, device="cuda")
    view = L.BowView()
    assert L.lib().plf_match_bow(m._h, C.byref(view), 3, C.c_float(0.7), 1, L.vp(z), 256, L.vp(z), None) == L.PLF_E_BADARG     # n_pairs > max_batch
    assert L.lib().plf_match_bow(m._h, C.byref(view), 1, C.c_float(0.7), 1, L.vp(z), 100000, L.vp(z), None) == L.PLF_E_BADARG  # stride > max_keypoints
    assert L.lib().plf_match_bow_kf(m._h, C.byref(view), 1, C.c_float(0.7), 1, L.vp(z), 256, L.vp(z), None) == L.PLF_E_BADARG  # f_has_mp missing
    m.close()



@SCHEDULES
@pytest.mark.parametrize("bands", [2, 5, 8, 13, 32])
def test_lines_speculative_bands(monkeypatch, bands, schedule):
    """banded speculative region growing (<= 8 frames in flight) with different band counts: same bits as the serial oracle"""
    _need_gpu()
    _set_schedule(monkeypatch, schedule)
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame, synth_batch
    monkeypatch.setenv("PLF_LSD_SPEC_BANDS", str(bands))
    ext = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=4)
    for seed in (6, 7, 8):
        _check(synth_frame(seed), 100, ext=ext)
    rng = np.random.default_rng(bands)
    img = (rng.integers(0, 256, (480, 640)) // 64 * 64).astype(np.uint8)          # blocky noise: thousands of tiny regions
    _check(img, 100, ext=ext)
    grad = np.add.outer(np.arange(480), np.arange(640)).astype(np.float64)
    _check(((np.sin(grad / 17.0) * 0.5 + 0.5) * 255).astype(np.uint8), 100, ext=ext)  # long diagonal regions crossing every band
    imgs = synth_batch(40 + bands, 4)
    res = ext.extract_batch(imgs)
    for f in range(4):
        ref = orc.line_extract(imgs[f], 100)
        assert np.array_equal(res[f][1], ref["desc"])
        assert np.array_equal(res[f][0]["startPointX"].view(np.uint32), ref["kl"]["startPointX"].view(np.uint32))
    ext.close()


@SCHEDULES
def test_lines_speculative_overflow_falls_back(monkeypatch, schedule):
    """record buffers too small: the commit kernel ignores the records and runs the serial loop itself"""
    _need_gpu()
    _set_schedule(monkeypatch, schedule)
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    monkeypatch.setenv("PLF_LSD_SPEC_RECCAP", "4")
    ext = LineSegment(nlines=100, max_width=640, max_height=480)
    _check(synth_frame(9), 100, ext=ext)
    ext.close()


def test_lines_speculative_slow_band_is_not_a_timeout(monkeypatch):
    """The commit wave of the one-launch schedule gives up (PLF_E_HIP) after a bounded number of polls WITHOUT a heartbeat of the band wave it waits for.
    With the bound cut to 7 ms a two-band VGA frame (each band wave works for far longer than that) still has to come out equal to the oracle: a band
    wave that keeps retiring seeds is slow, not missing.  (tools/soak.py seed 51591, a 3-level quantised checkerboard that takes 42 s per frame on the GPU
    and 2.4 s in the oracle, ran into the unconditional 7 s bound with 2 and 4 bands.)"""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    monkeypatch.setenv("PLF_LSD_SPEC_SPINS", "2048")
    monkeypatch.setenv("PLF_LSD_SPEC_BANDS", "2")
    ext = LineSegment(nlines=100, max_width=640, max_height=480)
    for seed in (9, 21):
        _check(synth_frame(seed), 100, ext=ext)
    ext.close()


@SCHEDULES
def test_lines_speculative_odd_widths(monkeypatch, schedule):
    """scaled widths that are not multiples of 8 / 32 (752 -> 602, 600 -> 480, 333 -> 266) through the speculative path"""
    _need_gpu()
    _set_schedule(monkeypatch, schedule)
    from rgbd_pl_slam_amd.synth import synth_frame
    for seed, (w, h) in enumerate([(752, 480), (600, 401), (333, 250)]):
        _check(synth_frame(60 + seed, w, h), 100)


@SCHEDULES
def test_lines_speculative_pathological_images(monkeypatch, schedule):
    """regions spanning every band, regions as long as the frame, and a log overflow caused by one giant region"""
    _need_gpu()
    _set_schedule(monkeypatch, schedule)
    from rgbd_pl_slam_amd import LineSegment
    y, x = np.mgrid[0:480, 0:640]
    imgs = [((x * 7) % 256).astype(np.uint8),                                   # vertical sawtooth: frame-high regions, one per period
            ((y * 9) % 256).astype(np.uint8),                                   # horizontal sawtooth: regions inside single bands
            ((x * 3 + y * 5) % 256).astype(np.uint8),                           # diagonal sawtooth
            (np.hypot(x - 320.0, y - 240.0) * 6 % 256).astype(np.uint8),        # rings: every angle, regions curving through all bands
            (((x // 16 + y // 16) % 2) * 200 + 20).astype(np.uint8),            # checkerboard: many short edges, corners
            np.clip(x * 0.6 + y * 0.4, 0, 255).astype(np.uint8)]                # gentle ramp: below the gradient threshold almost everywhere
    ext = LineSegment(nlines=100, max_width=640, max_height=480)
    for im in imgs:
        _check(np.ascontiguousarray(im), 100, ext=ext)
    ext.close()


@SCHEDULES
@pytest.mark.parametrize("halo", [0, 5, 40])
def test_lines_speculative_halo_rows(monkeypatch, halo, schedule):
    """warm-up rows above every band (default 16): any number of them, including none and more than a band is high, gives the serial result"""
    _need_gpu()
    _set_schedule(monkeypatch, schedule)
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    monkeypatch.setenv("PLF_LSD_SPEC_HALO", str(halo))
    ext = LineSegment(nlines=100, max_width=640, max_height=480)
    for seed in (12, 13):
        _check(synth_frame(seed), 100, ext=ext)
    ext.close()


def test_lines_thousands_of_rectangles(monkeypatch):
    """Textures of thousands of tiny regions: the rectangle capacity is the exact bound sw*sh/min_reg_size (a 4096-rectangle cap used to fail on the
    first image, found by tools/soak.py seed 246), and frames with more than 4096 segments sort through the global scratch row."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    # the soak's failing frame: hard-edged stripes, period ~ a few pixels, at an angle
    rng = np.random.default_rng(77000 + 246)
    rng.random(); rng.integers(0, 12)
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
    a = rng.uniform(0, np.pi); per = rng.uniform(3, 40)
    stripes = (127.5 + 120 * np.sign(np.sin((xx * np.cos(a) + yy * np.sin(a)) * 2 * np.pi / per))).astype(np.uint8)
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=2)
    ref = orc.line_extract(stripes, 100)
    kl, desc, eq = ls.ExtractLineSegment(stripes)
    assert kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"])
    yi, xi = np.mgrid[0:480, 0:640]
    checker = ((((yi // 10) + (xi // 10)) & 1) * 200).astype(np.uint8)
    res = ls.extract_batch(np.stack([checker, stripes]))                       # the same through the batched (speculative) path
    refc = orc.line_extract(checker, 100)
    assert res[0][0].tobytes() == refc["kl"].tobytes() and np.array_equal(res[0][1], refc["desc"])
    assert res[1][0].tobytes() == ref["kl"].tobytes() and np.array_equal(res[1][1], ref["desc"])
    # eight such frames: more rectangles than the batch's pooled NFA buffers -> the host-buffer call redoes the batch in halves by itself ...
    res = ls8 = None
    ls8 = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    res = ls8.extract_batch(np.stack([stripes] * 8))
    for f in range(8):
        assert res[f][0].tobytes() == ref["kl"].tobytes() and np.array_equal(res[f][1], ref["desc"])
    # ... and the fully device-resident call reports it.  (Up to 64 frames in flight take the one-wave-per-rectangle NFA kernel, which has no pooled buffers
    # and therefore no such limit: checked first.  PLF_NFA_FUSED=0 selects the staged kernels of the large batches for the rest.)
    import torch
    from rgbd_pl_slam_amd import _lib as L
    d_img = torch.from_numpy(np.stack([stripes] * 8)).cuda()
    d_lines = torch.zeros(8 * 100 * 68, dtype=torch.uint8, device="cuda"); d_desc = torch.zeros(8 * 100 * 32, dtype=torch.uint8, device="cuda")
    d_eq = torch.zeros(8 * 100 * 3, dtype=torch.float64, device="cuda"); d_n = torch.zeros(8, dtype=torch.int32, device="cuda")
    ls8.extract_batch_device(d_img, 640, 480, d_lines, d_desc, d_eq, d_n, 100)
    assert ls8.last_status() == L.PLF_OK
    for f in (0, 7):
        got = np.frombuffer(d_lines.cpu().numpy().tobytes(), L.KL_DTYPE)[f * 100:f * 100 + int(d_n[f])]
        assert got.tobytes() == ref["kl"].tobytes()
    ls8.tune("nfa_fused", 0)                                    # (the knobs are read from the environment when a handle is created; afterwards: plf_line_tune)
    # large-batch schedule: rectangles of fewer than 512 pixels never enter the pooled buffers (k_nfa_small), so these frames fit
    ls8.extract_batch_device(d_img, 640, 480, d_lines, d_desc, d_eq, d_n, 100)
    assert ls8.last_status() == L.PLF_OK
    for f in (0, 7):
        got = np.frombuffer(d_lines.cpu().numpy().tobytes(), L.KL_DTYPE)[f * 100:f * 100 + int(d_n[f])]
        assert got.tobytes() == ref["kl"].tobytes()
    ls8.tune("nfa_small", 0)                                    # every rectangle through the staged kernels
    res = ls8.extract_batch(np.stack([stripes] * 8))          # the pool overflows, the host-buffer call splits the batch
    for f in range(8):
        assert res[f][0].tobytes() == ref["kl"].tobytes() and np.array_equal(res[f][1], ref["desc"])
    assert ls8.last_status() == L.PLF_OK and not ls8.truncated(8).any()   # status and per-frame flags cover all the pieces of a batch that was redone in halves
    ls8.extract_batch_device(d_img, 640, 480, d_lines, d_desc, d_eq, d_n, 100)
    assert ls8.last_status() == L.PLF_E_RECTS
    ls8.extract_batch_device(d_img[:2], 640, 480, d_lines, d_desc, d_eq, d_n, 100)
    assert ls8.last_status() == L.PLF_OK
    got = np.frombuffer(d_lines.cpu().numpy().tobytes(), L.KL_DTYPE)[:int(d_n[0])]
    assert got.tobytes() == ref["kl"].tobytes()
    ls8.close()
    ls.close()
    # 1280x960 checkerboard: ~10k segments -> compaction flags and sort keys in global memory; top-400 and unsorted (all kept) outputs
    yi, xi = np.mgrid[0:960, 0:1280]
    big = ((((yi // 10) + (xi // 10)) & 1) * 200).astype(np.uint8)
    for nl in (400, 20000):
        ls = LineSegment(nlines=nl, max_width=1280, max_height=960)
        ref = orc.line_extract(big, nl)
        kl, desc, eq = ls.ExtractLineSegment(big)
        assert len(ref["kl"]) == (400 if nl == 400 else len(ls.segments(0))) and len(ls.segments(0)) > 8192
        assert kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"])
        assert np.allclose(eq, ref["eq"], rtol=0, atol=1e-9)
        ls.close()


def _is_prefix(segs, ref_seg):
    return len(segs) <= len(ref_seg) and np.array_equal(segs.view(np.uint32), ref_seg[:len(segs)].view(np.uint32))


def test_time_budget_on_the_pathological_soak_frame():
    """plf_line_params.max_ms (opt-in; the reference has no bound).  tools/soak.py seed 51591 -- hard three-level stripes -- keeps the serial LSD chain of ONE
    VGA frame busy for ~40 s on the GPU (2.4-5 s in the CPU oracle): with a 30 ms budget the call returns at once, reports PLF_W_TRUNCATED, and what it
    found is a prefix of the reference's segment list.  Without a budget the result is exact (checked on the same texture at 320x240 to keep the suite short)."""
    _need_gpu()
    import time
    from rgbd_pl_slam_amd import LineSegment, _lib as L
    from rgbd_pl_slam_amd.synth import texture_frame
    img, _ = texture_frame(51591)
    assert img.shape == (480, 640)
    ls = LineSegment(nlines=100, max_ms=30.0)
    ls.ExtractLineSegment(img)                       # first call: allocations of the speculative schedule
    t0 = time.perf_counter()
    kl, desc, eq = ls.ExtractLineSegment(img)
    dt = time.perf_counter() - t0
    assert dt < 1.5, "a 30 ms budget took %.2f s" % dt
    assert ls.last_status() == L.PLF_W_TRUNCATED and int(ls.truncated(1)[0]) == 1
    assert _is_prefix(ls.segments(0), orc.lsd_detect(img)["lines"])
    ls.close()
    small, _ = texture_frame(51591, size=(320, 240))
    _check(small, 100)                               # no budget: exact, however long it takes


def test_time_budget_leaves_a_prefix_and_is_invisible_when_not_spent():
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment, _lib as L
    from rgbd_pl_slam_amd.synth import synth_frame
    imgs = np.stack([synth_frame(9100 + i) for i in range(6)])
    refs = [orc.lsd_detect(im)["lines"] for im in imgs]
    # generous budget: nothing changes, nothing is reported
    ls = LineSegment(nlines=100, max_ms=20000.0)
    _check(imgs[0], 100, ext=ls)
    assert ls.last_status() == 0 and int(ls.truncated(1)[0]) == 0
    ls.close()
    # one frame (speculative schedule, one launch): 1 ms is a fraction of its region stage
    ls = LineSegment(nlines=100, max_ms=1.0)
    ls.ExtractLineSegment(imgs[0]); ls.ExtractLineSegment(imgs[0])
    segs = ls.segments(0)
    assert ls.last_status() == L.PLF_W_TRUNCATED and _is_prefix(segs, refs[0]) and len(segs) < len(refs[0])
    ls.close()
    # batches: 40 frames (speculative schedule, band waves and commit waves stop on their own clocks) and 700 (one wave per frame)
    for B, ms in ((40, 1.0), (700, 8.0)):
        ls = LineSegment(nlines=100, max_batch=B, max_ms=ms)
        batch = np.stack([imgs[i % 6] for i in range(B)])
        ls.extract_batch(batch); res = ls.extract_batch(batch)
        flags = ls.truncated(B)
        assert ls.last_status() == L.PLF_W_TRUNCATED and flags.sum() > 0
        for f in (0, 1, 5, B - 1):
            segs = ls.segments(f)
            assert _is_prefix(segs, refs[f % 6]), "batch %d frame %d" % (B, f)
            if not flags[f]:
                assert len(segs) == len(refs[f % 6])
        ls.close()


@pytest.mark.parametrize("B,ms", [(16, 2.0), (16, 4.0), (64, 3.0), (64, 8.0)])
def test_time_budget_two_launch_speculative_schedule(B, ms):
    """with 14-16 or more than 49 frames in flight the speculative schedule is two launches (band waves, then commit waves), each with a clock of its own.
    A band wave that runs out of time leaves an incomplete log; the commit wave -- which may well finish inside ITS budget -- must not take that log for the whole
    band.  Every frame is a prefix of the reference's segments, and a frame that is not flagged as truncated is complete."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment, _lib as L
    from rgbd_pl_slam_amd.synth import synth_frame
    imgs = np.stack([synth_frame(9200 + i) for i in range(8)])
    refs = [orc.lsd_detect(im)["lines"] for im in imgs]
    ls = LineSegment(nlines=100, max_batch=B, max_ms=ms)
    batch = np.stack([imgs[i % 8] for i in range(B)])
    ls.extract_batch(batch); ls.extract_batch(batch)
    flags = ls.truncated(B)
    status = ls.last_status()
    assert status in (0, L.PLF_W_TRUNCATED) and (status == L.PLF_W_TRUNCATED) == bool(flags.sum() > 0)
    for f in range(B):
        segs = ls.segments(f)
        assert _is_prefix(segs, refs[f % 8]), "frame %d" % f
        if not flags[f]:
            assert len(segs) == len(refs[f % 8]), "frame %d finished in time but holds %d of %d segments" % (f, len(segs), len(refs[f % 8]))
    ls.close()


def test_alternating_batch_sizes_share_the_speculation_buffers():
    """the band count of the speculative schedule depends on the batch size (48 / 32 / 16 / ...), and every change used to re-allocate its buffers.
    They are now indexed with the current band count as the stride and re-used whenever frames x bands fits the allocation: one handle, batch sizes in an order
    that shrinks and grows the band count (and switches the validation rounds on and off), every frame against the oracle."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    imgs = [synth_frame(8100 + i) for i in range(40)]
    refs = [orc.line_extract(im, 100) for im in imgs]
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=40)
    for B in (40, 1, 12, 3, 40, 2, 20, 1, 17, 8):
        off = (7 * B) % 40
        idx = [(off + i) % 40 for i in range(B)]
        res = ls.extract_batch(np.stack([imgs[i] for i in idx]))
        for f, i in enumerate(idx):
            assert res[f][0].tobytes() == refs[i]["kl"].tobytes() and np.array_equal(res[f][1], refs[i]["desc"]), (B, f)
    ls.close()


def test_one_handle_mixed_band_counts_and_batch_sizes():
    """the rows a band's records reach (spec_reach) lived behind the band counts at an offset of frames_cap * nbands with the band count of the CALL,
    which may exceed the band count of the allocation (the buffers are re-used whenever frames x bands fits): 16 frames x 8 bands allocate 128 slots, 2 frames x 64
    bands re-use them and indexed up to 16 * 64 + 2 * 2 * 64 ints of a 384-int buffer.  One handle, band counts and batch sizes mixed so that the later calls have
    MORE bands than the allocation, every frame against the oracle (the reach rows now have a buffer of their own, indexed by slot)."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    imgs = [synth_frame(8500 + i) for i in range(16)]
    refs = [orc.line_extract(im, 100) for im in imgs]
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=16)
    for (bands, B) in ((8, 16), (64, 2), (32, 4), (48, 1), (16, 8), (64, 1), (8, 16), (24, 5)):
        ls.tune("spec_bands", bands)
        idx = [(3 * bands + i) % 16 for i in range(B)]
        res = ls.extract_batch(np.stack([imgs[i] for i in idx]))
        for f, i in enumerate(idx):
            assert res[f][0].tobytes() == refs[i]["kl"].tobytes() and np.array_equal(res[f][1], refs[i]["desc"]), (bands, B, f)
    ls.close()


@pytest.mark.parametrize("knobs,B", [({"nfa_small": 0}, 3), ({"nfa_table": 0}, 2), ({"nfa_small": 1}, 5), ({"nfa_two_pass": 0, "nfa_fused": 4}, 9), ({"nfa_small": 1, "nfa_fused": 2}, 6)],
                         ids=["fused_with_table", "fused_no_table", "small_only_above_fused", "small_single_pass_plus_staged", "small_above_fused_two_pass"])
def test_lines_nfa_schedules(knobs, B):
    """Every NFA (rect_improve) schedule reachable through plf_line_tune gives the same bits (kernel index): k_nfa_fused (one wave per rectangle,
    with and without the value table), k_nfa_small in one pass followed by the staged kernels, k_nfa_small + k_nfa_small2 on a small batch.  The defaults are covered
    by every other test."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
    imgs = [synth_frame(8700 + i) if i % 2 == 0 else natural_frame(8700 + i) for i in range(B)]
    refs = [orc.line_extract(im, 100) for im in imgs]
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
    for k, v in knobs.items():
        ls.tune(k, v)
    res = ls.extract_batch(np.stack(imgs))
    for f in range(B):
        assert res[f][0].tobytes() == refs[f]["kl"].tobytes() and np.array_equal(res[f][1], refs[f]["desc"]), (knobs, f)
    kl, ld, eq = ls.ExtractLineSegment(imgs[0])
    assert kl.tobytes() == refs[0]["kl"].tobytes() and np.array_equal(ld, refs[0]["desc"])
    ls.close()


def test_slow_frame_warning_outputs_complete():
    """PLF_W_SLOW: a host-output call that takes far longer per frame than the handle's recent calls returns the warning -- the outputs are
    complete and exact -- and plf_line_last_status repeats it until the next call.  History of cheap frames (flat images: nothing to grow), then a window of the
    gravel photograph (regions that cross many bands: 13 validation rounds); the thresholds are lowered through plf_line_tune so that the test does not need a frame that takes seconds.  With
    slow_factor = 0 the same frame passes silently; a different image size starts a new history."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    import rgbd_pl_slam_amd._lib as L
    ls = LineSegment(nlines=100, max_width=640, max_height=480)
    ls.tune("slow_factor", 4.0); ls.tune("slow_floor_ms", 0.0)   # (4x: far above the jitter of a 1.6 ms call on a busy host, far below the ~15x of the hard frame)
    flat = np.full((480, 640), 90, np.uint8)
    for _ in range(10):
        ls.ExtractLineSegment(flat)
        assert L.last_warning == 0
    from rgbd_pl_slam_amd.synth import photo_frame
    noisy = photo_frame(51006)      # a window of the gravel photograph: ~25 ms with one frame in flight, 15x a flat frame (profiles/r06_photo_latency.txt)
    ref = orc.line_extract(noisy, 100)
    kl, ld, eq = ls.ExtractLineSegment(noisy)
    assert L.last_warning == L.PLF_W_SLOW
    assert ls.last_status() == L.PLF_W_SLOW
    assert kl.tobytes() == ref["kl"].tobytes() and np.array_equal(ld, ref["desc"])   # complete and exact
    ls.ExtractLineSegment(flat)
    assert L.last_warning == 0 and ls.last_status() == 0
    ls.tune("slow_factor", 0.0)
    ls.ExtractLineSegment(noisy)
    assert L.last_warning == 0
    ls.tune("slow_factor", 4.0)
    small = np.ascontiguousarray(noisy[:240, :320])
    ls.ExtractLineSegment(small)   # first call at another size: no history, no warning
    assert L.last_warning == 0
    ls.close()


def test_one_handle_alternating_image_sizes_rebuilds_the_nfa_table():
    """The table of NFA values (k_nfa_table) depends on the scaled image size (LOG_NT enters nfa_d's exit test): a handle that sees a different size must refill it.
    One handle, sizes 640x480 -> 320x240 -> 800x600 -> 640x480, single frames and a batch of 70 (the staged NFA kernels' hand-over), against the oracle."""
    _need_gpu()
    from rgbd_pl_slam_amd import LineSegment
    from rgbd_pl_slam_amd.synth import synth_frame
    ls = LineSegment(nlines=100, max_width=800, max_height=600, max_batch=70)
    for (w, h) in ((640, 480), (320, 240), (800, 600), (640, 480)):
        imgs = [synth_frame(8300 + i, w, h) for i in range(5)]
        refs = [orc.line_extract(im, 100) for im in imgs]
        kl, ld, eq = ls.ExtractLineSegment(imgs[0])
        assert kl.tobytes() == refs[0]["kl"].tobytes() and np.array_equal(ld, refs[0]["desc"]), (w, h)
        res = ls.extract_batch(np.stack([imgs[i % 5] for i in range(70)]))
        for f in range(70):
            assert res[f][0].tobytes() == refs[f % 5]["kl"].tobytes() and np.array_equal(res[f][1], refs[f % 5]["desc"]), (w, h, f)
    ls.close()
