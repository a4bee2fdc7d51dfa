"""GPU parity: HIP ORB extractor (through the C ABI) vs the CPU oracle, bit-exact.
Tolerances: NONE -- integer stages (pyramid, FAST scores/cells, octree, blur, BRIEF bits) must be equal;
float fields (x, y, size, angle, response) are compared on their bit patterns."""
import numpy as np
import pytest

import orc
import refgen
from conftest import gpu_available

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")


def _compare(img, nfeatures, stages=True, ext=None):
    from rgbd_pl_slam_amd import ORBextractor
    h, w = img.shape
    own = ext is None
    if own:
        ext = ORBextractor(nfeatures=nfeatures, max_width=w, max_height=h)
    kps, desc = ext(img)
    ref = orc.orb_extract(img, nfeatures=nfeatures, debug=stages)
    if stages:
        for l in range(8):
            assert np.array_equal(ext.pyramid_level(0, l), ref["pyr"][l]), "pyramid level %d" % l
            assert np.array_equal(ext.blurred_level(0, l), ref["blur"][l]), "blur level %d" % l
            cand = ext.candidates(0, l)
            assert len(cand) == ref["ncand"][l], "candidate count level %d: %d vs %d" % (l, len(cand), ref["ncand"][l])
    assert len(kps) == len(ref["kps"]), "keypoint count %d vs %d" % (len(kps), len(ref["kps"]))
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kps[f].view(np.uint32), ref["kps"][f].view(np.uint32)), f
    assert np.array_equal(kps["octave"], ref["kps"]["octave"])
    assert np.array_equal(desc, ref["desc"])
    if own:
        ext.close()
    return kps, desc


def test_orb_vga_synthetic_bit_exact():
    _need_gpu()
    from rgbd_pl_slam_amd.synth import synth_frame
    for seed in (0, 1):
        _compare(synth_frame(seed), 1000)


def test_orb_reference_fixture_images():
    _need_gpu()
    _compare(refgen.synth_image(9001, 640, 480), 1000)
    _compare(refgen.synth_image(9003, 320, 240), 500)


def test_orb_2000_features_and_odd_size():
    _need_gpu()
    from rgbd_pl_slam_amd.synth import synth_frame
    _compare(synth_frame(3), 2000)
    _compare(synth_frame(4, 752, 480), 1200)


def test_orb_1280x960_4000():
    _need_gpu()
    from rgbd_pl_slam_amd.synth import synth_frame
    _compare(synth_frame(5, 1280, 960), 4000, stages=False)


def test_orb_flat_and_noise_images():
    """edge cases: no corners at all (empty output), and dense noise (retry threshold, many ties)"""
    _need_gpu()
    from rgbd_pl_slam_amd import ORBextractor
    flat = np.full((480, 640), 128, np.uint8)
    ext = ORBextractor(max_width=640, max_height=480)
    kps, desc = ext(flat)
    assert len(kps) == 0 and desc.shape == (0, 32)
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    _compare(noise, 1000, ext=ext)
    low = (128 + rng.integers(-6, 7, (480, 640))).astype(np.uint8)   # only the minTh retry can fire
    _compare(low, 1000, ext=ext)
    ext.close()


@pytest.mark.parametrize("w,h", [(640, 480), (642, 481), (333, 250)])
def test_orb_blur_saturation_and_tail_columns(w, h):
    """The 7x7 blur's taps sum to 257 per axis (OpenCV 3.3's 8-bit fixed point, not renormalised): plateaus of 254 / 255 give column sums of 256 * 65536 and more, which
    SymmColumnVec_32s8u saturates to 255 -- the fp32 column pass of k_orb_level (round 6) relies on v_cvt_pk_u8_f32 saturating there -- and exact halves (x.5 * 65536)
    round to even in the vector columns and up in the w % 4 tail columns.  Bright plateaus, a 254 / 255 checkerboard, ramps that hit the ties, odd sizes."""
    _need_gpu()
    rng = np.random.default_rng(w)
    img = np.full((h, w), 255, np.uint8)
    img[: h // 3] = 254
    yy, xx = np.mgrid[0:h, 0:w]
    img[h // 3: h // 2] = np.where(((yy[h // 3: h // 2] // 9) + (xx[h // 3: h // 2] // 9)) % 2 == 0, 255, 253).astype(np.uint8)
    img[h // 2: 2 * h // 3] = (255 - (xx[h // 2: 2 * h // 3] % 64)).astype(np.uint8)
    img[2 * h // 3:] = rng.choice(np.array([0, 128, 254, 255], np.uint8), size=(h - 2 * h // 3, w))
    for _ in range(30):   # dark rectangles: corners on the plateaus
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        img[y0:y0 + int(rng.integers(8, 40)), x0:x0 + int(rng.integers(8, 40))] = int(rng.integers(0, 256))
    _compare(img, 1000)


def test_orb_batch_equals_single():
    _need_gpu()
    from rgbd_pl_slam_amd import ORBextractor
    from rgbd_pl_slam_amd.synth import synth_batch
    imgs = synth_batch(10, 4)
    ext = ORBextractor(max_width=640, max_height=480, max_batch=4)
    res = ext.extract_batch(imgs)
    for f in range(4):
        ref = orc.orb_extract(imgs[f])
        assert np.array_equal(res[f][1], ref["desc"])
        assert np.array_equal(res[f][0]["angle"].view(np.uint32), ref["kps"]["angle"].view(np.uint32))
    ext.close()


def test_orb_errors():
    _need_gpu()
    from rgbd_pl_slam_amd import ORBextractor, PlfError
    ext = ORBextractor(max_width=640, max_height=480)
    with pytest.raises(PlfError):
        ext(np.zeros((960, 1280), np.uint8))        # larger than the handle was created for
    with pytest.raises(PlfError):
        ORBextractor(max_width=40, max_height=40)   # the reference divides by zero on such sizes
    ext.close()


def test_gpu_equals_reference_operator_fixture():
    """HIP extractor vs tests/golden/ref_glue_operator.json: the output of the reference binary's own
    ORBextractor::operator() (so@0x76da0) on the same seeded images, key points and descriptors bit for bit."""
    import json, os
    _need_gpu()
    from rgbd_pl_slam_amd import ORBextractor
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glue_operator.json")) as fh:
        cases = json.load(fh)["cases"]
    for c in cases:
        img = refgen.synth_image(c["seed"], c["w"], c["h"])
        ext = ORBextractor(nfeatures=c["nfeatures"], max_width=c["w"], max_height=c["h"])
        kps, desc = ext(img)
        ref = np.array(c["kp"], np.uint32).view(np.float32).reshape(-1, 5)
        assert len(kps) == c["n"], c["w"]
        for j, f in enumerate(("x", "y", "size", "angle", "response")):
            assert np.array_equal(kps[f].view(np.uint32), ref[:, j].view(np.uint32)), (c["w"], f)
        assert np.array_equal(kps["octave"], np.array(c["octave"], np.int32)), c["w"]
        assert np.array_equal(desc, np.frombuffer(bytes.fromhex(c["desc"]), np.uint8).reshape(-1, 32)), c["w"]


def test_get_tables_equals_reference_ctor_fixture():
    """a1: plf_orb_get_tables (C ABI, HIP library) against the tables the reference constructor itself produced
    (ORBextractor::ORBextractor so@0x73050 executed by oracle/refprobe -> tests/golden/ref_orb_tables.json), bit patterns compared"""
    _need_gpu()
    import json
    import os
    import struct
    from rgbd_pl_slam_amd import ORBextractor, PlfError
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_orb_tables.json")))["cases"]
    done = 0
    for c in cases:
        nl = int(c["nlevels"])
        sf = struct.unpack("f", struct.pack("I", c["scaleFactor_bits"]))[0]
        try:
            ext = ORBextractor(nfeatures=int(c["nfeatures"]), scaleFactor=sf, nlevels=nl, max_width=1280, max_height=960)
        except PlfError:
            continue    # parameter sets outside the handle's geometry limits (levels, quota) are covered by the oracle test
        bits = lambda a: np.asarray(a, np.float32).view(np.uint32).tolist()
        assert ext.GetLevels() == nl
        assert bits(ext.GetScaleFactors()) == c["scale"]
        assert bits(ext.GetInverseScaleFactors()) == c["inv"]
        assert bits(ext.GetScaleSigmaSquares()) == c["sigma2"]
        assert bits(ext.GetInverseScaleSigmaSquares()) == c["invsigma2"]
        assert ext.GetFeaturesPerLevel().tolist() == c["perLevel"]
        ext.close()
        done += 1
    assert done >= 3, "only %d of %d reference parameter sets fit a handle" % (done, len(cases))


@pytest.mark.parametrize("w,h,sf,nlev,ini,mn,nf", [(1280, 960, 1.1, 8, 26, 5, 2000), (1280, 960, 1.1, 3, 38, 6, 3000), (723, 542, 1.3, 5, 20, 7, 800)])
def test_level_geometries_with_clamped_last_cells(w, h, sf, nlev, ini, mn, nf):
    """regressions found by tools/soak.py in the fused level kernel: a level whose LAST cell row / column is cut at the border so that the cell
    before it has a shorter computed region than hCell / wCell (and the last one none), and a last tile that is a single 4-column group wide
    (e.g. the 1164x873 and 723x542 levels of a 1280x960 pyramid at scale 1.1); every level compared stage by stage"""
    _need_gpu()
    from rgbd_pl_slam_amd import ORBextractor
    from rgbd_pl_slam_amd.synth import synth_frame
    img = synth_frame(31, w, h)
    ref = orc.orb_extract(img, nfeatures=nf, scale_factor=sf, nlevels=nlev, ini_th=ini, min_th=mn, debug=True)
    e = ORBextractor(nfeatures=nf, scaleFactor=sf, nlevels=nlev, iniThFAST=ini, minThFAST=mn, max_width=w, max_height=h)
    kps, desc = e(img)
    for l in range(nlev):
        assert np.array_equal(e.pyramid_level(0, l), ref["pyr"][l]), "pyramid level %d" % l
        assert np.array_equal(e.blurred_level(0, l), ref["blur"][l]), "blurred level %d" % l
        assert len(e.candidates(0, l)) == ref["ncand"][l], "FAST candidates of level %d" % l
    assert kps.tobytes() == ref["kps"].tobytes() and np.array_equal(desc, ref["desc"])
    e.close()
