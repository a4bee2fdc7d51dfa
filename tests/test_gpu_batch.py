"""GPU parity of the multi-GPU batch driver (plf_batch_*, rgbd_pl_slam_amd/csrc/batch_host.hip): host frames in, features (and matches
against a replicated local map) out, through per-GPU worker threads with pinned double-buffered staging.  Every frame must equal the CPU
oracle's output for that frame, whatever chunk / slot / GPU it went through.  BASELINE configs 3 and 4 run here as specified."""
import numpy as np
import pytest

import matchgen
import orc
from conftest import gpu_available

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")


def _same_orb(got, ref, tag):
    assert len(got["kps"]) == len(ref["kps"]), "%s: %d vs %d key points" % (tag, len(got["kps"]), len(ref["kps"]))
    for name in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(got["kps"][name].view(np.uint32), ref["kps"][name].view(np.uint32)), "%s: key point field %s" % (tag, name)
    assert np.array_equal(got["desc"], ref["desc"]), "%s: ORB descriptors" % tag


def _same_lines(got, ref, tag):
    assert len(got["lines"]) == len(ref["kl"]), "%s: %d vs %d lines" % (tag, len(got["lines"]), len(ref["kl"]))
    for name in got["lines"].dtype.names:
        assert np.array_equal(got["lines"][name].view(np.uint32), ref["kl"][name].view(np.uint32)), "%s: KeyLine field %s" % (tag, name)
    assert np.array_equal(got["ldesc"], ref["desc"]), "%s: LBD descriptors" % tag
    assert np.allclose(got["line_eq"], ref["eq"], rtol=0, atol=1e-4), "%s: line equations" % tag


def test_config4_batch_of_8_1280x960_4000_400():
    """BASELINE configs[3] as one GPU of the 8 sees it: 8 frames of 1280x960 in flight, 4000 ORB features + 400 lines."""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd.synth import synth_frame
    imgs = np.stack([synth_frame(900 + i, 1280, 960) for i in range(8)])
    bx = BatchExtractor(nfeatures=4000, nlines=400, width=1280, height=960, frames_in_flight=8, devices=[0])
    res = bx.extract(imgs)
    for f in range(8):
        _same_orb(res[f], orc.orb_extract(imgs[f], nfeatures=4000), "frame %d" % f)
        _same_lines(res[f], orc.line_extract(imgs[f], 400), "frame %d" % f)
        assert len(res[f]["kps"]) >= 3900 and len(res[f]["lines"]) == 400
    bx.close()


def test_config3_batch_vga_2000_200_with_local_map_and_ragged_chunks():
    """BASELINE configs[2]: VGA, 2000 + 200, 8 frames in flight; 21 frames = two full chunks + a ragged one, so both pipeline slots are
    reused; matches against a replicated local map (SearchByProjection for points and lines) ride along."""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd.synth import synth_frame
    n = 21
    imgs = np.stack([synth_frame(300 + i) for i in range(n)])
    r0 = orc.orb_extract(imgs[0], nfeatures=2000); l0 = orc.line_extract(imgs[0], 200)
    mp = matchgen.make_local_map(r0["kps"], r0["desc"], 3000, 5)
    ml = matchgen.make_map_lines(l0["kl"], l0["desc"], 400, 6)
    scale = orc.orb_tables(2000, 1.2, 8)["scale"]
    bounds = (0.0, 0.0, 640.0, 480.0)
    bx = BatchExtractor(nfeatures=2000, nlines=200, width=640, height=480, frames_in_flight=8, devices=[0], max_mappoints=4096, max_maplines=512)
    bx.set_local_map(mp, ml, th=3.0, nnratio=0.8, bounds=bounds)
    for rep in range(2):   # second call: slots and handles reused across calls
        res = bx.extract(imgs)
        for f in range(n):
            ro = orc.orb_extract(imgs[f], nfeatures=2000); rl = orc.line_extract(imgs[f], 200)
            _same_orb(res[f], ro, "frame %d" % f); _same_lines(res[f], rl, "frame %d" % f)
            rm, rn = orc.search_by_projection_map(ro["kps"], ro["desc"], None, scale, bounds, mp, 3.0, 0.8, np.full(len(ro["kps"]), -1, np.int32))
            assert res[f]["n_kp_matches"] == rn and np.array_equal(res[f]["match_of_kp"], rm), "frame %d: point matches" % f
            lm, ln = orc.search_lines_by_projection(rl["kl"], rl["desc"], scale, ml, 3.0, 0.8, np.full(len(rl["kl"]), -1, np.int32))
            assert res[f]["n_line_matches"] == ln and np.array_equal(res[f]["match_of_line"], lm), "frame %d: line matches" % f
            if f == 0:
                assert rn > 100 and ln > 10
    t = bx.last_timing()
    assert t["total"] > 0
    bx.close()


def test_padded_pinned_and_rgb_inputs():
    """row pitch > width, frame stride > pitch * height, pinned caller memory (no staging copy) and RGB / BGR frames
    (Tracking::GrabImageRGBD colour conversion on the device) all give the features of the plain gray frames"""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor, FMT_RGB8, FMT_BGR8, pinned_array, free_pinned
    from rgbd_pl_slam_amd.synth import synth_frame
    n, w, h = 5, 320, 240
    gray = np.stack([synth_frame(40 + i, w, h) for i in range(n)])
    ref = [(orc.orb_extract(g, nfeatures=500), orc.line_extract(g, 60)) for g in gray]
    bx = BatchExtractor(nfeatures=500, nlines=60, width=w, height=h, frames_in_flight=2, devices=[0])
    padded = np.zeros((n, h + 3, w + 24), np.uint8); padded[:, :h, :w] = gray
    for tag, arr in (("padded", padded[:, :h, :w]), ("tight", gray)):
        res = bx.extract(arr)
        for f in range(n):
            _same_orb(res[f], ref[f][0], tag); _same_lines(res[f], ref[f][1], tag)
    pin = pinned_array((n, h, w)); pin[:] = gray
    res = bx.extract(pin)
    for f in range(n):
        _same_orb(res[f], ref[f][0], "pinned"); _same_lines(res[f], ref[f][1], "pinned")
    free_pinned(pin)
    bx.close()
    rng = np.random.default_rng(7)
    rgb = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    rgb[..., 1] = gray    # structured green channel so that features exist
    for fmt, bgr in ((FMT_RGB8, 0), (FMT_BGR8, 1)):
        bc = BatchExtractor(nfeatures=500, nlines=60, width=w, height=h, frames_in_flight=2, devices=[0], input_format=fmt)
        res = bc.extract(rgb)
        for f in range(n):
            g = orc.rgb_to_gray(rgb[f], bgr)
            _same_orb(res[f], orc.orb_extract(g, nfeatures=500), "rgb"); _same_lines(res[f], orc.line_extract(g, 60), "rgb")
        bc.close()


def test_rgbd_frame_constructor_batch():
    """plf_batch_extract_rgbd = the RGB-D Frame constructor (include/Frame.h:60) for a batch of host frames: colour -> gray, uint16 depth ->
    float, ExtractORB / ExtractLSD, UndistortKeyPoints, ComputeStereoFromRGBD and the line-side members, then SearchByProjection on mvKeysUn /
    mvuRight / mvKeylinesUn.  7 frames with 3 in flight (ragged last chunk), padded depth pitch; also without depth (monocular rule)."""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor, FMT_BGR8
    from rgbd_pl_slam_amd.frame import camera, TUM1
    from rgbd_pl_slam_amd.synth import synth_frame
    n, w, h = 7, 640, 480
    rng = np.random.default_rng(11)
    gray = np.stack([synth_frame(500 + i, w, h) for i in range(n)])
    bgr = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8); bgr[..., 1] = gray
    dpad = np.zeros((n, h + 2, w + 10), np.uint16)
    dpad[:, :h, :w] = rng.integers(2000, 30000, (n, h, w), dtype=np.uint16)
    dpad[:, :h, :w][rng.uniform(0, 1, (n, h, w)) < 0.15] = 0          # holes: no depth
    d16 = dpad[:, :h, :w]
    cam = camera(**TUM1)
    c9 = np.array([TUM1[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")], np.float32)
    factor = np.float32(1.0 / 5000.0)
    g0 = orc.rgb_to_gray(bgr[0], 1)
    r0 = orc.orb_extract(g0, nfeatures=1000); l0 = orc.line_extract(g0, 100)
    un0, ur0, _ = orc.frame_tail(r0["kps"], orc.depth_to_float(np.ascontiguousarray(d16[0]), factor), c9, TUM1["bf"])
    mp = matchgen.make_local_map(un0, r0["desc"], 3000, 5, uright=ur0)
    lun0 = orc.line_tail(l0["kl"], None, c9, TUM1["bf"])[0]
    ml = matchgen.make_map_lines(lun0, l0["desc"], 400, 6)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    bounds = (-10.0, -12.0, 655.0, 490.0)      # Frame::ComputeImageBounds of a distorted camera reach outside the image
    bx = BatchExtractor(nfeatures=1000, nlines=100, width=w, height=h, frames_in_flight=3, devices=[0], input_format=FMT_BGR8,
                        max_mappoints=4096, max_maplines=512, rgbd=True)
    bx.set_local_map(mp, ml, th=3.0, nnratio=0.8, bounds=bounds)
    for with_depth in (True, False):
        res = bx.extract(bgr, depth=d16 if with_depth else None, cam=cam, depth_factor=float(factor))
        for f in range(n):
            g = orc.rgb_to_gray(bgr[f], 1)
            ro = orc.orb_extract(g, nfeatures=1000); rl = orc.line_extract(g, 100)
            _same_orb(res[f], ro, "frame %d" % f); _same_lines(res[f], rl, "frame %d" % f)
            tag = "frame %d depth %d" % (f, with_depth)
            if with_depth:
                df = orc.depth_to_float(np.ascontiguousarray(d16[f]), factor)
                un, ur, kd = orc.frame_tail(ro["kps"], df, c9, TUM1["bf"])
                assert np.array_equal(res[f]["uright"].view(np.uint32), ur.view(np.uint32)), tag
                assert np.array_equal(res[f]["kp_depth"].view(np.uint32), kd.view(np.uint32)), tag
                assert (ur > 0).sum() > 500
            else:
                df = None
                un = orc.frame_tail(ro["kps"], np.zeros((h, w), np.float32), c9, TUM1["bf"])[0]
                ur = None
                assert np.all(res[f]["uright"] == -1) and np.all(res[f]["kp_depth"] == -1), tag
            for name in un.dtype.names:
                assert np.array_equal(res[f]["kps_un"][name].view(np.uint32), un[name].view(np.uint32)), "%s: mvKeysUn.%s" % (tag, name)
            lun, urs, ure, ds, de = orc.line_tail(rl["kl"], df, c9, TUM1["bf"])
            for name in lun.dtype.names:
                assert np.array_equal(res[f]["lines_un"][name].view(np.uint32), lun[name].view(np.uint32)), "%s: mvKeylinesUn.%s" % (tag, name)
            for got, exp in ((res[f]["uright_start"], urs), (res[f]["uright_end"], ure), (res[f]["depth_start"], ds), (res[f]["depth_end"], de)):
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), tag
            rm, rn = orc.search_by_projection_map(un, ro["desc"], ur, scale, bounds, mp, 3.0, 0.8, np.full(len(un), -1, np.int32))
            assert res[f]["n_kp_matches"] == rn and np.array_equal(res[f]["match_of_kp"], rm), "%s: point matches" % tag
            lm, ln = orc.search_lines_by_projection(lun, rl["desc"], scale, ml, 3.0, 0.8, np.full(len(lun), -1, np.int32))
            assert res[f]["n_line_matches"] == ln and np.array_equal(res[f]["match_of_line"], lm), "%s: line matches" % tag
            if f == 0:
                assert rn > 100 and ln > 10
    # the RGB-D entry needs the buffers of plf_batch_params.rgbd
    plain = BatchExtractor(nfeatures=500, nlines=50, width=w, height=h, frames_in_flight=2, devices=[0])
    with pytest.raises(RuntimeError):
        plain.extract(gray[:1], cam=cam)
    plain.close()
    bx.close()


def test_chunk_redo_when_the_rectangle_pool_overflows(monkeypatch):
    """Eight frames of hard-edged stripes hold more LSD rectangles than the pooled NFA buffers of an 8-frame batch (PLF_E_RECTS): the worker redoes
    the chunk through the splitting entry point and repeats the Frame tail and the line matching on the fresh lines; a normal chunk follows in the
    same call (slot reuse after a redo).  (PLF_NFA_FUSED=0: the staged NFA kernels of the large batches -- up to 64 frames in flight normally take the
    one-wave-per-rectangle kernel, which has no pool.)"""
    _need_gpu()
    monkeypatch.setenv("PLF_NFA_FUSED", "0")
    monkeypatch.setenv("PLF_NFA_SMALL", "0")      # (with k_nfa_small only rectangles of 512 pixels or more use the pool: these frames would fit)
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd.frame import camera, TUM1
    from rgbd_pl_slam_amd.synth import synth_frame
    rng = np.random.default_rng(77000 + 246)
    rng.random(); rng.integers(0, 12)
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
    a = rng.uniform(0, np.pi); per = rng.uniform(3, 40)
    stripes = (127.5 + 120 * np.sign(np.sin((xx * np.cos(a) + yy * np.sin(a)) * 2 * np.pi / per))).astype(np.uint8)
    imgs = np.stack([stripes] * 8 + [synth_frame(600 + i) for i in range(3)])
    n = len(imgs)
    d16 = np.random.default_rng(5).integers(2000, 30000, (n, 480, 640), dtype=np.uint16)
    cam = camera(**TUM1)
    c9 = np.array([TUM1[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")], np.float32)
    refs = {}
    for f in (0, 8, 9, 10):
        refs[f] = orc.line_extract(imgs[f], 100)
    for f in range(1, 8):
        refs[f] = refs[0]
    lun0 = orc.line_tail(refs[8]["kl"], None, c9, TUM1["bf"])[0]
    ml = matchgen.make_map_lines(lun0, refs[8]["desc"], 300, 6)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    bx = BatchExtractor(nfeatures=1000, nlines=100, width=640, height=480, frames_in_flight=8, devices=[0], max_mappoints=16, max_maplines=512, rgbd=True)
    bx.set_local_map(None, ml, th=3.0, nnratio=0.8, bounds=(0.0, 0.0, 640.0, 480.0))
    res = bx.extract(imgs, depth=d16, cam=cam)
    for f in range(n):
        _same_lines(res[f], refs[f], "frame %d" % f)
        df = orc.depth_to_float(np.ascontiguousarray(d16[f]), np.float32(1.0 / 5000.0))
        lun, urs, ure, ds, de = orc.line_tail(refs[f]["kl"], df, c9, TUM1["bf"])
        for name in lun.dtype.names:
            assert np.array_equal(res[f]["lines_un"][name].view(np.uint32), lun[name].view(np.uint32)), "frame %d: mvKeylinesUn.%s" % (f, name)
        assert np.array_equal(res[f]["depth_end"].view(np.uint32), de.view(np.uint32)), "frame %d" % f
        lm, ln = orc.search_lines_by_projection(lun, refs[f]["desc"], scale, ml, 3.0, 0.8, np.full(len(lun), -1, np.int32))
        assert res[f]["n_line_matches"] == ln and np.array_equal(res[f]["match_of_line"], lm), "frame %d: line matches" % f
    assert res[8]["n_line_matches"] > 10
    bx.close()


def test_all_visible_gpus_share_one_batch():
    """n_devices = 0: every visible GPU gets a contiguous block (plf_batch_shard); on the 1-GPU test box this is one worker, on an
    8-GPU node the same call exercises eight -- the per-frame outputs do not depend on the partition"""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor, shard
    from rgbd_pl_slam_amd.synth import synth_frame
    n = 11
    imgs = np.stack([synth_frame(70 + i, 320, 240) for i in range(n)])
    bx = BatchExtractor(nfeatures=500, nlines=50, width=320, height=240, frames_in_flight=4)
    assert bx.n_devices >= 1 and bx.devices == list(range(bx.n_devices))
    blocks = [shard(n, bx.n_devices, d) for d in range(bx.n_devices)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    res = bx.extract(imgs)
    for f in range(n):
        _same_orb(res[f], orc.orb_extract(imgs[f], nfeatures=500), "frame %d" % f)
        _same_lines(res[f], orc.line_extract(imgs[f], 50), "frame %d" % f)
    bx.close()


def _check_frames(res, imgs, nfeat, nlines, frames=None, local_map=None, tag=""):
    for f in (range(len(imgs)) if frames is None else frames):
        ro = orc.orb_extract(imgs[f], nfeatures=nfeat); rl = orc.line_extract(imgs[f], nlines)
        _same_orb(res[f], ro, "%sframe %d" % (tag, f)); _same_lines(res[f], rl, "%sframe %d" % (tag, f))
        if local_map is not None:
            mp, ml, scale, bounds = local_map
            rm, rn = orc.search_by_projection_map(ro["kps"], ro["desc"], None, scale, bounds, mp, 3.0, 0.8, np.full(len(ro["kps"]), -1, np.int32))
            assert res[f]["n_kp_matches"] == rn and np.array_equal(res[f]["match_of_kp"], rm), "%sframe %d: point matches" % (tag, f)
            lm, ln = orc.search_lines_by_projection(rl["kl"], rl["desc"], scale, ml, 3.0, 0.8, np.full(len(rl["kl"]), -1, np.int32))
            assert res[f]["n_line_matches"] == ln and np.array_equal(res[f]["match_of_line"], lm), "%sframe %d: line matches" % (tag, f)


def test_two_workers_on_one_gpu_37_vga_frames_with_local_map():
    """More than one worker for real: devices = [0, 0] gives two worker threads -- each with its own handles, streams, pinned slots and
    local-map replica -- sharing the one GPU of the test box, exactly the code an 8-GPU node runs with eight.  37 VGA frames = blocks of 18 and 19
    (plf_batch_shard), 8 in flight: each worker runs two full chunks and a ragged one while the other is active; both write into ONE set of caller
    arrays.  Two calls in a row (hand-off post / wait_done twice, slots reused).  Every frame equals the oracle."""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor, shard
    from rgbd_pl_slam_amd.synth import synth_frame
    n = 37
    imgs = np.stack([synth_frame(1300 + i) for i in range(n)])
    r0 = orc.orb_extract(imgs[0], nfeatures=1000); l0 = orc.line_extract(imgs[0], 100)
    mp = matchgen.make_local_map(r0["kps"], r0["desc"], 3000, 5)
    ml = matchgen.make_map_lines(l0["kl"], l0["desc"], 400, 6)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    bounds = (0.0, 0.0, 640.0, 480.0)
    bx = BatchExtractor(nfeatures=1000, nlines=100, width=640, height=480, frames_in_flight=8, devices=[0, 0], max_mappoints=4096, max_maplines=512)
    assert bx.n_devices == 2 and bx.devices == [0, 0]
    assert [shard(n, 2, g) for g in range(2)] == [(0, 18), (18, 37)]
    bx.set_local_map(mp, ml, th=3.0, nnratio=0.8, bounds=bounds)
    for rep in range(2):
        res = bx.extract(imgs)
        _check_frames(res, imgs, 1000, 100, local_map=(mp, ml, scale, bounds), tag="call %d " % rep)
    # a new local map replaces the replicas of BOTH workers
    mp2 = matchgen.make_local_map(r0["kps"], r0["desc"], 2000, 15)
    bx.set_local_map(mp2, ml, th=3.0, nnratio=0.8, bounds=bounds)
    res = bx.extract(imgs)
    _check_frames(res, imgs, 1000, 100, frames=(0, 17, 18, 36), local_map=(mp2, ml, scale, bounds), tag="new map ")
    bx.close()


def test_eight_workers_config4_exactly_as_specified():
    """BASELINE configs[3] exactly as written: batch 64 of 1280x960 frames, 4000 ORB + 400 lines, sharded over EIGHT workers -- eight NUMA-bound
    threads, eight sets of pinned slots, eight stream sets -- with 8 frames in flight each; without an 8-GPU node all eight drive device 0.  Every worker must get
    plf_batch_shard(64, 8, r) = 8 frames; a sample of frames from every worker against the oracle, the rest against the frames that repeat them."""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor, shard
    from rgbd_pl_slam_amd.synth import synth_frame
    distinct = [synth_frame(2300 + i, 1280, 960) for i in range(16)]
    order = [(5 * i) % 16 for i in range(64)]                 # every worker's block holds 8 different images
    imgs = np.stack([distinct[k] for k in order])
    bx = BatchExtractor(nfeatures=4000, nlines=400, width=1280, height=960, frames_in_flight=8, devices=[0] * 8)
    assert bx.n_devices == 8 and [shard(64, 8, r) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    res = bx.extract(imgs)
    refs = {}
    for f in list(range(0, 64, 9)) + [7, 8, 63]:              # frames 0, 9, 18, ...: one or two per worker, both ends of a block
        k = order[f]
        if k not in refs:
            refs[k] = (orc.orb_extract(distinct[k], nfeatures=4000), orc.line_extract(distinct[k], 400))
        _same_orb(res[f], refs[k][0], "frame %d" % f)
        _same_lines(res[f], refs[k][1], "frame %d" % f)
    first = {}
    for f in range(64):                                        # every frame equals the first frame that carried the same image (different workers, different slots)
        k = order[f]
        if k in first:
            g = res[first[k]]
            assert res[f]["kps"].tobytes() == g["kps"].tobytes() and np.array_equal(res[f]["desc"], g["desc"]), f
            assert res[f]["lines"].tobytes() == g["lines"].tobytes() and np.array_equal(res[f]["ldesc"], g["ldesc"]), f
        else:
            first[k] = f
    tms = [bx.worker_timing(w) for w in range(8)]
    assert all(t["total"] > 0 and t["gpu_wait"] >= 0 for t in tms)
    tot = bx.last_timing()
    assert tot["total"] == max(t["total"] for t in tms)
    bx.close()


def test_four_workers_on_one_gpu_config4_shape():
    """BASELINE configs[3] shape (1280x960, 4000 ORB + 400 lines) on FOUR workers (devices = [0, 0, 0, 0]): 16 frames = 4 per worker, 2 in flight, so every
    worker pipelines two chunks through both slots while three others compete for the GPU"""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd.synth import synth_frame
    n = 16
    imgs = np.stack([synth_frame(2100 + i, 1280, 960) for i in range(n)])
    bx = BatchExtractor(nfeatures=4000, nlines=400, width=1280, height=960, frames_in_flight=2, devices=[0, 0, 0, 0])
    assert bx.n_devices == 4
    for rep in range(2):
        res = bx.extract(imgs)
        _check_frames(res, imgs, 4000, 400, frames=None if rep == 0 else (0, 3, 4, 8, 15), tag="call %d " % rep)
    # fewer frames than workers: blocks of 0 / 1 frames (an idle worker must not touch the outputs or block the call)
    res = bx.extract(imgs[:3])
    _check_frames(res, imgs[:3], 4000, 400, tag="3 frames ")
    bx.close()


def test_rectangle_pool_redo_in_one_worker_while_the_other_runs(monkeypatch):
    """PLF_E_RECTS redo (host-memory re-extraction in halves + line matching on the fresh lines) inside worker 0 while worker 1 processes normal frames on the
    same GPU; lines-only batch with map lines -- the mvScaleFactors table no longer depends on an ORB handle"""
    _need_gpu()
    monkeypatch.setenv("PLF_NFA_FUSED", "0")      # the staged NFA kernels (pooled buffers): what chunks of more than 64 frames take
    monkeypatch.setenv("PLF_NFA_SMALL", "0")      # every rectangle through them, so that the pool overflows
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd.synth import synth_frame
    rng = np.random.default_rng(77000 + 246)
    rng.random(); rng.integers(0, 12)
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
    a = rng.uniform(0, np.pi); per = rng.uniform(3, 40)
    stripes = (127.5 + 120 * np.sign(np.sin((xx * np.cos(a) + yy * np.sin(a)) * 2 * np.pi / per))).astype(np.uint8)
    normal = [synth_frame(2600 + i) for i in range(11)]
    imgs = np.stack([stripes] * 8 + normal[:1] + normal)          # worker 0: frames 0-9 (8 stripes + 2 normal), worker 1: frames 10-19
    n = len(imgs)
    assert n == 20
    refs = {}
    for f in range(n):
        refs[f] = refs[0] if 0 < f < 8 else orc.line_extract(imgs[f], 100)
    ml = matchgen.make_map_lines(refs[8]["kl"], refs[8]["desc"], 300, 6)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    for nfeat in (1000, 0):
        bx = BatchExtractor(nfeatures=nfeat, nlines=100, width=640, height=480, frames_in_flight=8, devices=[0, 0], max_mappoints=16, max_maplines=512)
        bx.set_local_map(None, ml, th=3.0, nnratio=0.8, bounds=(0.0, 0.0, 640.0, 480.0))
        res = bx.extract(imgs)
        for f in range(n):
            _same_lines(res[f], refs[f], "nfeatures %d frame %d" % (nfeat, f))
            lm, ln = orc.search_lines_by_projection(refs[f]["kl"], refs[f]["desc"], scale, ml, 3.0, 0.8, np.full(len(refs[f]["kl"]), -1, np.int32))
            assert res[f]["n_line_matches"] == ln and np.array_equal(res[f]["match_of_line"], lm), "nfeatures %d frame %d: line matches" % (nfeat, f)
        assert res[8]["n_line_matches"] > 10
        if nfeat:
            for f in (8, 19):
                _same_orb(res[f], orc.orb_extract(imgs[f], nfeatures=nfeat), "frame %d" % f)
        bx.close()


def test_match_counts_follow_the_callers_capacity():
    """kp_capacity / line_capacity below what the frame yields: PLF_E_CAPACITY, the rows hold the first `capacity` features and n_*_matches counts
    the matches of THOSE"""
    _need_gpu()
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd import _lib as L
    from rgbd_pl_slam_amd.synth import synth_frame
    imgs = np.stack([synth_frame(310 + i) for i in range(3)])
    r0 = orc.orb_extract(imgs[0], nfeatures=1000); l0 = orc.line_extract(imgs[0], 100)
    mp = matchgen.make_local_map(r0["kps"], r0["desc"], 2000, 5)
    ml = matchgen.make_map_lines(l0["kl"], l0["desc"], 300, 6)
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    bounds = (0.0, 0.0, 640.0, 480.0)
    bx = BatchExtractor(nfeatures=1000, nlines=100, width=640, height=480, frames_in_flight=4, devices=[0], max_mappoints=4096, max_maplines=512)
    bx.set_local_map(mp, ml, th=3.0, nnratio=0.8, bounds=bounds)
    bx.kp_capacity = 300; bx.nlines = 40          # what the wrapper hands to the driver as the caller's capacities
    out = bx.alloc_outputs(3)
    st = bx.extract_into(imgs, out)
    assert st == L.PLF_E_CAPACITY
    for f in range(3):
        ro = orc.orb_extract(imgs[f], nfeatures=1000); rl = orc.line_extract(imgs[f], 100)
        assert len(ro["kps"]) > 300 and len(rl["kl"]) > 40
        assert out["n_kps"][f] == 300 and out["n_lines"][f] == 40
        assert out["kps"][f].tobytes() == ro["kps"][:300].tobytes() and out["lines"][f].tobytes() == rl["kl"][:40].tobytes()
        rm, rn = orc.search_by_projection_map(ro["kps"], ro["desc"], None, scale, bounds, mp, 3.0, 0.8, np.full(len(ro["kps"]), -1, np.int32))
        assert np.array_equal(out["match_of_kp"][f], rm[:300]) and out["n_kp_matches"][f] == int((rm[:300] >= 0).sum())
        lm, ln = orc.search_lines_by_projection(rl["kl"], rl["desc"], scale, ml, 3.0, 0.8, np.full(len(rl["kl"]), -1, np.int32))
        assert np.array_equal(out["match_of_line"][f], lm[:40]) and out["n_line_matches"][f] == int((lm[:40] >= 0).sum())
        if f == 0:
            assert rn > int((rm[:300] >= 0).sum()) > 0    # the cut really drops matches
    bx.close()


def test_empty_and_bad_arguments():
    _need_gpu()
    import ctypes as C
    from rgbd_pl_slam_amd import _lib as L
    from rgbd_pl_slam_amd.batch import BatchExtractor
    bx = BatchExtractor(nfeatures=500, nlines=50, width=320, height=240, frames_in_flight=4, devices=[0])
    out = bx.alloc_outputs(1)
    img = np.zeros((1, 240, 320), np.uint8)
    assert bx.extract_into(img, out) == 0 and out["n_kps"][0] == 0 and out["n_lines"][0] == 0     # flat image: no features, no error
    big = np.zeros((1, 480, 640), np.uint8)
    with pytest.raises(L.PlfError) as e:
        bx.extract_into(big, bx.alloc_outputs(1))
    assert e.value.status == L.PLF_E_BADARG
    bx.close()


def test_batch_workers_are_bound_to_the_numa_node_of_their_gpu(monkeypatch):
    """plf_batch_worker_affinity: a worker thread is bound to the CPUs of its GPU's NUMA node before it allocates its pinned slots (two workers on GPU 0 land on
    the same node); hosts without the sysfs view -- or PLF_BATCH_NO_AFFINITY=1 -- leave the thread unbound, and the call still works"""
    _need_gpu()
    import os
    from rgbd_pl_slam_amd.batch import BatchExtractor
    from rgbd_pl_slam_amd import PlfError
    bx = BatchExtractor(nfeatures=500, nlines=50, width=320, height=240, frames_in_flight=2, devices=[0, 0])
    a0, a1 = bx.worker_affinity(0), bx.worker_affinity(1)
    assert a0 == a1 and a0[0] >= -1 and a0[1] >= 0
    if a0[0] >= 0:   # the node is known: the thread was bound to (a subset of) its CPUs
        cpus = open("/sys/devices/system/node/node%d/cpulist" % a0[0]).read().strip()
        assert a0[1] > 0, "node %d (CPUs %s) known, but the worker was not bound" % (a0[0], cpus)
        assert a0[1] <= len(os.sched_getaffinity(0))
    with pytest.raises(PlfError):
        bx.worker_affinity(2)
    bx.close()
    monkeypatch.setenv("PLF_BATCH_NO_AFFINITY", "1")
    bx = BatchExtractor(nfeatures=500, nlines=50, width=320, height=240, frames_in_flight=2, devices=[0])
    assert bx.worker_affinity(0)[1] == 0
    bx.close()
